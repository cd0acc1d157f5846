"""Forward + backward through autograd (training use), 128x3x512x512 fp32, J=3 db4 symmetric."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, n=20):
    for _ in range(40): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


x = torch.randn(128, 3, 512, 512, device=dev, requires_grad=True)
xfm = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
ifm = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
res = {}
for fused in (True, False):
    lowlevel.FUSED_LEVELS = fused
    def fb():
        yl, yh = xfm(x)
        rec = ifm((yl, yh))
        g, = torch.autograd.grad(rec, x, torch.ones_like(rec))
        return g
    def f_only():
        with torch.no_grad():
            yl, yh = xfm(x)
            return ifm((yl, yh))
    res['fused' if fused else 'per_level'] = {'fwd_inv_then_backward_ms': timed(fb), 'fwd_inv_ms': timed(f_only)}
print(json.dumps(res))
