"""Per-level timings of config 5 (db8 periodization float16, 32x16 planes): strip kernel (forced) against the tile kernel at
every level's plane size.  usage: python tools/gpu_cfg5_levels.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel as _ll

dev = 'cuda:0'
a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)


def timeit(fn, n=20):
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


xfm = pw.DWTForward(J=1, wave='db8', mode='periodization').to(dev).half()
ifm = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half()
for W in (2048, 1024, 512, 256):
    x = torch.randn(32, 16, W, W, device=dev).half()
    with torch.no_grad():
        c = xfm(x)
        out = []
        for force in (False, True):
            ops.STREAM_FORCE = force
            tf = timeit(lambda: xfm(x)); kf = pw.last_kernel()
            ti = timeit(lambda: ifm(c)); ki = pw.last_kernel()
            out.append((tf, ti, kf, ki))
        ops.STREAM_FORCE = False
    by = 4 * x.numel()
    print('W=%4d  fwd policy %.4f ms [%s]  forced %.4f ms [%s] (%.0f GB/s)   inv policy %.4f [%s]  forced %.4f [%s]' % (
        W, out[0][0], out[0][2], out[1][0], out[1][2], by / out[1][0] / 1e6, out[0][1], out[0][3], out[1][1], out[1][3]), flush=True)
