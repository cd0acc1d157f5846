"""QMF (lowpass banks only) against two-bank variants of the one-level strip kernels, same box: analysis and synthesis,
12-20 taps, float32 / float16.  usage: python tools/gpu_qmf_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel as _ll

dev = 'cuda:0'
_ll.FUSED_LEVELS = False
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


for wave, mode, shape, dtype in (('db6', 'symmetric', (64, 3, 1024, 1024), torch.float32), ('db7', 'symmetric', (64, 3, 1024, 1024), torch.float32),
                                 ('db8', 'symmetric', (64, 3, 1024, 1024), torch.float32), ('db10', 'symmetric', (64, 3, 1024, 1024), torch.float32),
                                 ('db7', 'periodization', (16, 16, 2048, 2048), torch.float16), ('db8', 'periodization', (16, 16, 2048, 2048), torch.float16)):
    x = torch.randn(*shape, device=dev).to(dtype)
    xfm = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev).to(dtype)
    ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dtype)
    with torch.no_grad():
        c = xfm(x)
        out = []
        for q in (True, False):
            if not q:
                xfm._qmf = lambda *b: False
                ifm._qmf = lambda *b: False
            tf = timeit(lambda: xfm(x)); kf = pw.last_kernel()
            ti = timeit(lambda: ifm(c)); ki = pw.last_kernel()
            out.append((tf, ti, kf, ki))
    print('%-5s %-13s %s %s  fwd qmf %.4f / two-bank %.4f ms   inv qmf %.4f / two-bank %.4f ms   [%s | %s]' % (
        wave, mode, shape, str(dtype)[6:], out[0][0], out[1][0], out[0][1], out[1][1], out[0][2], out[0][3]), flush=True)
