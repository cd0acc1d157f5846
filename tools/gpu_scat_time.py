"""ScatLayer forward (inference and training mode) and backward at config 4's shape and at 512 columns.
usage: python tools/gpu_scat_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw

dev = 'cuda:0'
a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        for _ in range(20):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return sorted(res)[1]


for shape in ((256, 3, 256, 256), (255, 3, 256, 256), (64, 3, 512, 512)):
    x = torch.randn(*shape, device=dev)
    sl = pw.ScatLayer().to(dev)
    with torch.no_grad():
        tf = timeit(lambda: sl(x)); kf = pw.last_kernel()
    xg = x.clone().requires_grad_(True)
    tt = timeit(lambda: sl(xg)); kt = pw.last_kernel()
    z = sl(xg)
    gz = torch.randn_like(z)
    tb = timeit(lambda: torch.autograd.grad(z, xg, gz, retain_graph=True)); kb = pw.last_kernel()
    px = x.numel()
    print('%s %s  fwd %.4f ms (%.3f of 11 B/px) [%s]   fwd(train) %.4f ms (%.3f of 23 B/px) [%s]   bwd %.4f ms (%.3f of 23 B/px) [%s]' % (
        os.environ.get('WL_LIB', 'product'), shape, tf, 11 * px / tf / 1e6 / 8000, kf, tt, 23 * px / tt / 1e6 / 8000, kt, tb, 23 * px / tb / 1e6 / 8000, kb), flush=True)
