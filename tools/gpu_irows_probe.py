"""Streaming synthesis kernel: parity against the per-level tile kernels and HIP-event timings (whole planes /
automatic cutting / every plane cut), next to the per-level path."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


def per_level(yl, yh, tg, mode):
    ll = yl
    for h in reversed(yh):
        if ll.shape[-2] > h.shape[-2]: ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]: ll = ll[..., :-1]
        ll = ops.sfb2d(ll, h, *tg, mode)
    return ll


res = {}
for wave in os.environ.get('PROBE_WAVES', 'db4').split(','):
    h0, h1 = filters.dwt_analysis_taps(wave)
    g0, g1 = filters.dwt_synthesis_taps(wave)
    th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
    tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
    for planes in [int(v) for v in os.environ.get('SWEEP_PLANES', '384').split(',')]:
        x = torch.randn(planes, 1, 512, 512, device=dev)
        for J in (3, 1):
            yl, yh = x, []
            for _ in range(J):
                yl, hi = ops.afb2d(yl, *th, 1)
                yh.append(hi)
            want = per_level(yl, yh, tg, 1)
            key = '%s_p%d_J%d' % (wave, planes, J)
            res[key + '_tile'] = timed(lambda: per_level(yl, yh, tg, 1))
            for strips in (1, 0, 2):
                got = ops.sfb2d_fused(yl, yh, *tg, 1, strips=strips)
                if got is None:
                    res[key + '_s%d' % strips] = None
                    continue
                res[key + '_s%d_err' % strips] = float((got - want).abs().max() / want.abs().max())
                res[key + '_s%d_rec' % strips] = float((got - x).abs().max())
                res[key + '_s%d' % strips] = timed(lambda: ops.sfb2d_fused(yl, yh, *tg, 1, strips=strips))
print(json.dumps(res))
