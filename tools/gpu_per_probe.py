"""Periodization (one level per streaming launch) against the per-level tile kernels, forward, 128x3x512x512 fp32."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel
from pytorch_wavelets_amd import _lib
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, n=30):
    for _ in range(60): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


res = {}
with torch.no_grad():
    x = torch.randn(128, 3, 512, 512, device=dev)
    for wave in ('haar', 'db3', 'db5', 'db4'):
        for mode in ('periodization', 'periodic'):
            xfm = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev)
            r = {}
            for fused in (True, False):
                lowlevel.FUSED_LEVELS = fused
                a = xfm(x)
                r['fused' if fused else 'tile'] = timed(lambda: xfm(x))
                r['k_' + ('fused' if fused else 'tile')] = _lib.get().wl_last_kernel().decode()[-40:]
            lowlevel.FUSED_LEVELS = True
            res['%s_%s' % (wave, mode[:6])] = r
print(json.dumps(res))
