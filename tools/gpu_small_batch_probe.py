"""Few planes: the streaming kernels (whole planes / every plane cut in two) against the per-level tile kernels."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0, h1 = filters.dwt_analysis_taps('db4'); g0, g1 = filters.dwt_synthesis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]


def timed(f, n=30):
    for _ in range(60): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


def fwd_tile(x, J):
    ll, yh = x, []
    for _ in range(J):
        ll, h = ops.afb2d(ll, *th, 1); yh.append(h)
    return ll, yh


def inv_tile(yl, yh):
    ll = yl
    for h in reversed(yh):
        if ll.shape[-2] > h.shape[-2]: ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]: ll = ll[..., :-1]
        ll = ops.sfb2d(ll, h, *tg, 1)
    return ll


res = {}
for planes in [int(v) for v in os.environ.get('SWEEP_PLANES', '32,64,96,128,192,256').split(',')]:
    x = torch.randn(planes, 1, 512, 512, device=dev)
    yl, yh = fwd_tile(x, 3)
    r = {'fwd_tile': timed(lambda: fwd_tile(x, 3)), 'inv_tile': timed(lambda: inv_tile(yl, yh))}
    for s in (0, 1, 2):
        r['fwd_s%d' % s] = timed(lambda: ops.afb2d_fused(x, *th, 1, 3, strips=s))
        r['inv_s%d' % s] = timed(lambda: ops.sfb2d_fused(yl, yh, *tg, 1, strips=s))
    res['p%d' % planes] = r
print(json.dumps(res))
