"""How long does the GPU take to reach its steady state?  Consecutive timed blocks of the same forward transform."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
with torch.no_grad():
    x = torch.randn(128, 3, 512, 512, device=dev)
    xfm = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
    ifm = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
    yl, yh = xfm(x)
    torch.cuda.synchronize()
    out = {}
    for name, f in (('fwd', lambda: xfm(x)), ('inv', lambda: ifm((yl, yh))), ('fwd_again', lambda: xfm(x))):
        blocks = []
        for b in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(25): f()
            e1.record(); torch.cuda.synchronize()
            blocks.append(round(e0.elapsed_time(e1) / 25, 4))
        out[name] = blocks
print(json.dumps(out))
