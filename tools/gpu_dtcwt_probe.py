"""DTCWT J=3 (near_sym_a / qshift_a) on 64x3x512x512: forward, inverse and the level-1 inverse alone (HIP events)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


res = {'lib': os.environ.get('WL_LIB')}
with torch.no_grad():
    x = torch.randn(64, 3, 512, 512, device=dev)
    for J in (1, 3):
        fx, ix = pw.DTCWTForward(J=J).to(dev), pw.DTCWTInverse().to(dev)
        yl, yh = fx(x)
        rec = ix((yl, yh))
        res['J%d_err' % J] = float((rec - x).abs().max())
        res['J%d_fwd' % J] = timed(lambda: fx(x))
        res['J%d_inv' % J] = timed(lambda: ix((yl, yh)))
print(json.dumps(res))
