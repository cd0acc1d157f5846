"""Random shapes on the real GPU for the round-4 kernels against the paths they replace: the small-plane DWT kernels
(WlAfbSmall / WlSfbSmall) against the per-level tile kernels, SWTForward's level kernel (WlSwtLevel) and the rotationally
symmetric level 1 (WlDtFwd1Rot) against the single-axis kernels.  Prints the failures (none expected) and a summary line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_wavelets_amd import ops, filters
from pytorch_wavelets_amd.dwt import lowlevel as dwl
from pytorch_wavelets_amd.dtcwt import lowlevel as dl
from pytorch_wavelets_amd.dtcwt import transform_funcs as tf
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
took = {'afb_small': 0, 'sfb_small': 0, 'swt': 0, 'rot': 0}
waves = ['haar', 'db2', 'db3', 'db4', 'db5', 'db6', 'db8', 'db10', 'bior2.2', 'sym4']
h0o, _, h1o, _, h2o, _ = filters.biort('near_sym_b_bp')
hrot = [dl.prep_filt(v, 1).to(dev) for v in (h0o, h1o, h2o)]
for seed in range(n):
    rng = np.random.RandomState(7000 + seed)
    wave = waves[rng.randint(len(waves))]
    h0, h1 = filters.dwt_analysis_taps(wave); g0, g1 = filters.dwt_synthesis_taps(wave)
    L = len(h0)
    mode = ['zero', 'symmetric', 'reflect', 'periodic', 'periodization'][rng.randint(5)]
    mi = {'zero': 0, 'symmetric': 1, 'reflect': 4, 'periodic': 6, 'periodization': 2}[mode]
    dt = torch.float16 if rng.rand() < 0.25 else torch.float32
    tol = 4e-3 if dt == torch.float16 else 3e-5
    th = [torch.tensor(np.asarray(v), dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
    tg = [torch.tensor(np.asarray(v), dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
    # ---- small planes
    J = int(rng.randint(1, 5))
    H, W = int(rng.randint(2, 72)), int(rng.randint(2, 72))
    if rng.rand() < 0.4:
        H = W = [8, 16, 32, 64][rng.randint(4)]
    planes = int(rng.randint(1, 700))
    x = torch.randn(planes, 1, H, W, device=dev).to(dt)
    res = ops.afb2d_small(x, *th, mi, J)
    if res is not None:
        took['afb_small'] += 1
        ll, yh = x, []
        for _ in range(J):
            ll, h = ops.afb2d(ll, *th, mi); yh.append(h)
        for a, b in zip([res[0]] + list(res[1]), [ll] + yh):
            e = float((a.float() - b.float()).abs().max() / max(1.0, float(b.float().abs().max())))
            if a.shape != b.shape or not e < tol:
                bad += 1; print('BAD afb_small', seed, wave, mode, J, H, W, planes, dt, e)
        if L % 2 == 0:
            got = ops.sfb2d_small(ll, yh, *tg, mi)
            if got is not None:
                took['sfb_small'] += 1
                rec = ll
                for h in reversed(yh):
                    r = rec
                    if r.shape[-2] > h.shape[-2]: r = r[..., :-1, :]
                    if r.shape[-1] > h.shape[-1]: r = r[..., :-1]
                    rec = ops.sfb2d(r, h, *tg, mi)
                e = float((got.float() - rec.float()).abs().max() / max(1.0, float(rec.float().abs().max())))
                if got.shape != rec.shape or not e < tol:
                    bad += 1; print('BAD sfb_small', seed, wave, mode, J, H, W, planes, dt, e)
    # ---- one level of the stationary transform
    if mode != 'periodization':
        d = int(rng.randint(1, 5))
        N, C, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 4)), int(rng.randint(3, 200)), int(rng.randint(3, 300))
        xb = torch.randn(N, 4 * C, H, W, device=dev).to(dt)
        xs = xb[:, 0::4] if rng.rand() < 0.5 else xb[:, :C].contiguous()
        y = ops.swt2d_level(xs, th[0], th[1], th[2], th[3], d, dwl._ATROUS_EXT[mode])
        if y is not None:
            took['swt'] += 1
            dwl.FUSED_LEVELS = False
            try:
                y2 = dwl.afb2d_atrous(xs, tuple(th), mode, d)
            finally:
                dwl.FUSED_LEVELS = True
            e = float((y.float() - y2.float()).abs().max() / max(1.0, float(y2.float().abs().max())))
            if y.shape != y2.shape or not e < tol:
                bad += 1; print('BAD swt', seed, wave, mode, d, N, C, H, W, dt, e)
    # ---- level 1 with the band-pass diagonal
    if seed % 3 == 0:
        N, C, H, W = int(rng.randint(1, 5)), int(rng.randint(1, 5)), 2 * int(rng.randint(1, 120)), 2 * int(rng.randint(1, 160))
        x = torch.randn(N, C, H, W, device=dev).to(dt)
        m = 'symmetric' if rng.rand() < 0.6 else 'zero'
        hr = [v.to(torch.float32) for v in hrot]
        got = tf.fwd_j1_rot(x, *hr, False, 1, m)
        tf.FUSED_ROT = False
        try:
            old = tf.fwd_j1_rot(x, *hr, False, 1, m)
        finally:
            tf.FUSED_ROT = True
        took['rot'] += 1
        for a, b in zip(got, old):
            e = float((a.float() - b.float()).abs().max() / max(1.0, float(b.float().abs().max())))
            if a.shape != b.shape or not e < 2 * tol:
                bad += 1; print('BAD rot', seed, m, N, C, H, W, dt, e)
torch.cuda.synchronize()
print(json.dumps({'cases': n, 'runs': took, 'bad': bad}))
