"""Time the fused DTCWT inverse (levels 2 + 1, wl_dtcwt_inv_level21) alone: events around a loop of launches on a pre-filled
stream.  WL_LIB selects an A/B build.  usage: python tools/gpu_dti21_time.py [N C H W]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops

shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 3, 512, 512)
dev = 'cuda:0'
torch.manual_seed(0)
x = torch.randn(*shape, device=dev)
xfm, ifm = pw.DTCWTForward(J=2).to(dev), pw.DTCWTInverse().to(dev)
yl, yh = xfm(x)
a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)


def run():
    return ops.dtcwt_inv21(yl, yh[1], yh[0], ifm.g0o, ifm.g1o, ifm.g0a, ifm.g0b, ifm.g1a, ifm.g1b, 1)


y = run()
assert y is not None
err = float((y - x).abs().max())
res = []
for rep in range(5):
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    for _ in range(20):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50)
res.sort()
by = 20 * x.numel()
print('%-22s %s  median %.4f ms  min %.4f  (%.0f GB/s of 20 B/px)  roundtrip err %.2e  %s' % (
    os.environ.get('WL_LIB', 'product'), shape, res[2], res[0], by / res[2] / 1e6, err, pw.last_kernel()))
