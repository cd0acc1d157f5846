"""Random shapes on the real GPU for the round-5 lattice kernels (csrc/wl_lattice.h: WlAfbStrip / WlSfbStrip <.., QMF = 1, LAT = 1>)
against the two-bank strip kernels they stand in for: 12-20 tap orthogonal wavelets, every mode, float32 / float16, random
plane counts and sizes, one level forward and inverse through the strip entry points of the C ABI (forced).  A fifth of the
cases hands the hinted launch banks that are NOT an orthogonal pair (a random lowpass with its exact mirror, or a perturbed
table): the device must reject the lattice and the armed two-bank variant must do the work - the result still has to equal the
two-bank kernel's.  Prints the failures (none expected) and a summary line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
waves = ['db6', 'db7', 'db8', 'db10', 'sym6', 'sym7', 'sym8', 'sym10', 'coif2', 'coif3']
modes = {'zero': 0, 'symmetric': 1, 'reflect': 4, 'periodic': 6, 'periodization': 2}
bad = lat = rej = 0
for seed in range(n):
    rng = np.random.RandomState(9000 + seed)
    wave = waves[rng.randint(len(waves))]
    h0, h1 = (np.asarray(v, dtype=np.float64) for v in filters.dwt_analysis_taps(wave))
    g0, g1 = (np.asarray(v, dtype=np.float64) for v in filters.dwt_synthesis_taps(wave))
    L = len(h0)
    sign = np.array([1.0, -1.0] * (L // 2))
    kind = rng.randint(5)
    if kind == 0:          # a random lowpass with its exact mirror: the host's hint passes, the pair is not orthogonal
        h0 = rng.randn(L) / 3; h1 = sign * h0[::-1]; g0 = rng.randn(L) / 3; g1 = sign * g0[::-1]
    mode = list(modes)[rng.randint(5)]
    mi = modes[mode]
    dt = torch.float16 if rng.rand() < 0.4 else torch.float32
    tol = 3e-3 if dt == torch.float16 else 3e-6
    planes = int(rng.randint(1, 7))
    H = int(rng.randint(2 * L, 300)); W = 4 * int(rng.randint((2 * L + 3) // 4 + 1, 360))
    if mi == 2 and H % 2:
        H += 1
    x = torch.randn(1, planes, H, W, device=dev).to(dt)
    th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
    tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
    want_rej = kind == 0 or (dt == torch.float16 and False)
    out = {}
    for hinted in (False, True):
        with ops.qmf_hint(hinted):
            c0 = pw.launch_count()
            a = ops.afb2d_stream(x, *th, mi, force=True)
            ka = pw.kernels_since(c0)
            if a is None:
                break
            c0 = pw.launch_count()
            r = ops.sfb2d_stream(a[0], a[1], *tg, mi, force=True)
            kr = pw.kernels_since(c0)
        out[hinted] = (a, r, ka, kr)
    if len(out) < 2:
        continue
    (a0, r0, _, _), (a1, r1, ka, kr) = out[False], out[True]
    if L != 18:
        assert any('WlTapPrep' in k for k in ka) and any(k.rstrip('>').endswith(', 1, 1') for k in ka if 'WlAfbStrip' in k), ka
    lat += 1
    rej += kind == 0
    def err(p, q):
        return float((p.float() - q.float()).abs().max() / max(1e-6, float(q.float().abs().max())))
    es = [err(a1[0], a0[0]), err(a1[1], a0[1])] + ([err(r1, r0)] if r0 is not None and r1 is not None else [])
    if not max(es) < tol:
        bad += 1
        print('BAD', seed, wave, mode, dt, planes, H, W, 'kind', kind, es, ka, kr)
print('round-5 lattice fuzz: %d cases, %d through the hinted launches (%d with banks the device must reject), %d mismatches' % (n, lat, rej, bad))
