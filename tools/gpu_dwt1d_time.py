"""DWT1DForward J=3 db4 symmetric 64x16x65536 float32 (the f3 row of bench.py's other_configs) and a few other shapes:
fused kernel against the per-level path.  usage: python tools/gpu_dwt1d_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel as _ll

dev = 'cuda:0'
a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)


def timeit(fn, n=30):
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


for wave, mode, J, shape, dtype in (('db4', 'symmetric', 3, (64, 16, 65536), torch.float32), ('db4', 'symmetric', 3, (64, 16, 65536), torch.float16),
                                    ('db8', 'zero', 4, (64, 16, 65536), torch.float32), ('db2', 'periodization', 2, (4096, 16, 1024), torch.float32),
                                    ('db4', 'symmetric', 1, (64, 16, 65536), torch.float32)):
    x = torch.randn(*shape, device=dev).to(dtype)
    xfm = pw.DWT1DForward(J=J, wave=wave, mode=mode).to(dev).to(dtype)
    with torch.no_grad():
        tf = timeit(lambda: xfm(x)); kf = pw.last_kernel()
        _ll.FUSED_LEVELS = False
        tp = timeit(lambda: xfm(x)); kp = pw.last_kernel()
        _ll.FUSED_LEVELS = True
    by = 2 * x.numel() * x.element_size()
    print('%-5s %-13s J=%d %s %s  fused %.4f ms (%.0f GB/s, %.3f of 8 TB/s)   per level %.4f ms   [%s | %s]' % (
        wave, mode, J, shape, str(dtype)[6:], tf, by / tf / 1e6, by / tf / 1e6 / 8000, tp, kf, kp), flush=True)
