"""Random shapes / wavelets / modes / depths / dtypes on the real GPU: the streaming kernels (whole planes and cut
planes) against the per-level tile kernels.  Prints the failures (none expected) and a summary line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = took_a = took_s = 0
for seed in range(n):
    rng = np.random.RandomState(seed)
    wave = ['haar', 'db2', 'db3', 'db4', 'db5', 'db6'][rng.randint(6)]
    h0, h1 = filters.dwt_analysis_taps(wave); g0, g1 = filters.dwt_synthesis_taps(wave)
    L = len(h0)
    mode = ['zero', 'symmetric', 'reflect', 'periodic'][rng.randint(4)]
    mi = {'zero': 0, 'symmetric': 1, 'reflect': 4, 'periodic': 6}[mode]
    J = int(rng.randint(1, 4))
    lo = max(2, L) * 2 ** (J - 1) + 2
    H = int(rng.randint(lo, lo + 300)); W = int(rng.randint(lo, lo + [60, 200, 700, 1100][rng.randint(4)]))
    if rng.rand() < 0.5:
        W = (W + 3) // 4 * 4
    planes = int(rng.randint(1, 40))
    dt = torch.float16 if rng.rand() < 0.25 else torch.float32
    tol = 4e-3 if dt == torch.float16 else 2e-5
    x = torch.randn(planes, 1, H, W, device=dev).to(dt)
    th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
    tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
    ll, yh = x, []
    for _ in range(J):
        ll, h = ops.afb2d(ll, *th, mi); yh.append(h)
    rec = ll
    for h in reversed(yh):
        r = rec
        if r.shape[-2] > h.shape[-2]: r = r[..., :-1, :]
        if r.shape[-1] > h.shape[-1]: r = r[..., :-1]
        rec = ops.sfb2d(r, h, *tg, mi)
    for strips in (1, 2):
        res = ops.afb2d_fused(x, *th, mi, J, strips=strips) if not (mode == 'periodic' and J > 1) else None
        if res is not None:
            took_a += 1
            for a, b in zip([res[0]] + list(res[1]), [ll] + yh):
                e = float((a.float() - b.float()).abs().max() / b.float().abs().max())
                if a.shape != b.shape or not e < tol:
                    bad += 1; print('BAD analysis', seed, wave, mode, J, H, W, planes, dt, strips, e)
        got = ops.sfb2d_fused(ll, yh, *tg, mi, strips=strips)
        if got is not None:
            took_s += 1
            e = float((got.float() - rec.float()).abs().max() / rec.float().abs().max())
            if got.shape != rec.shape or not e < tol:
                bad += 1; print('BAD synthesis', seed, wave, mode, J, H, W, planes, dt, strips, e)
torch.cuda.synchronize()
print(json.dumps({'cases': n, 'analysis_runs': took_a, 'synthesis_runs': took_s, 'bad': bad}))
