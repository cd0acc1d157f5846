"""SWTForward timings: the one-launch-per-level kernel (wl_swt2d_level) against the single-axis path, and what it moves
(5 plane sizes per level: x in, four sub-bands out).  usage: python tools/gpu_swt_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel as _ll
from pytorch_wavelets_amd.dwt.transform2d import SWTForward

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


for wave, mode, J, shape, dtype in (('db2', 'periodic', 2, (16, 3, 512, 512), torch.float32), ('db2', 'periodic', 2, (64, 3, 512, 512), torch.float32),
                                    ('db4', 'symmetric', 3, (64, 3, 256, 256), torch.float32), ('db2', 'periodic', 2, (64, 3, 512, 512), torch.float16)):
    x = torch.randn(*shape, device=dev).to(dtype)
    m = SWTForward(J=J, wave=wave, mode=mode).to(dev).to(dtype)
    with torch.no_grad():
        c0 = pw.launch_count(); m(x); k1 = pw.kernels_since(c0)
        t1 = timeit(lambda: m(x))
        _ll.FUSED_LEVELS = False
        c0 = pw.launch_count(); m(x); k0 = pw.kernels_since(c0)
        t0 = timeit(lambda: m(x))
        _ll.FUSED_LEVELS = True
    by = J * 5 * x.numel() * x.element_size()
    print('%s %s J=%d %s %s: level kernel %.4f ms = %.0f GB/s (%.3f of 8 TB/s) [%s]   single-axis path %.4f ms [%d launches]' % (
        wave, mode, J, shape, str(dtype).split('.')[-1], t1, by / t1 / 1e6, by / t1 / 1e6 / 8000, k1[0], t0, len(k0)), flush=True)
