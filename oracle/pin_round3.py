"""Round-3 fixtures from the REAL reference (merged into tests/golden/index.json, the other fixtures are left alone):
  dwt_r3_*      deep periodization pyramids on small images: levels of 1-4 samples under 12-20 taps, where the reference's
                roll() (dwt/lowlevel.py:9-25) degenerates into the identity (shift >= twice the length) - forward, inverse
                and both hand-written backward passes; the oracle is pinned on them to 1e-10 first
  ext_dwt1d_r3_* 1-D periodization of signals shorter than the filter (DWT1DForward / DWT1DInverse, dwt/transform1d.py)
  ext_afb1d_per function-level afb1d / sfb1d in mode 'periodization' with ODD tap counts (3, 5, 7) and short signals

    PYTHONPATH=tools/ref_shim:/root/reference python oracle/pin_round3.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import wavelet_oracle as wo             # noqa: E402
import pytorch_wavelets as pw                       # noqa: E402  (the reference)
from pytorch_wavelets.dwt import lowlevel as rdl    # noqa: E402

torch.set_default_dtype(torch.float64)
GOLD = os.path.join(ROOT, 'tests', 'golden')
idx_path = os.path.join(GOLD, 'index.json')
index = json.load(open(idx_path))
rng = np.random.RandomState(333)
TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def npy(t):
    return t.detach().numpy()


def save(name, meta, **arrs):
    np.savez_compressed(os.path.join(GOLD, name + '.npz'),
                        **{k: (np.asarray(v).astype(np.float32) if np.asarray(v).dtype == np.float64 else np.asarray(v))
                           for k, v in arrs.items()})
    index[name] = meta


def f32(a):
    return a.astype(np.float32).astype(np.float64)


# ---- deep periodization pyramids (2-D)
for ci, (wave, J, shape) in enumerate([('db6', 5, (1, 2, 50, 28)), ('db7', 5, (2, 1, 50, 28)), ('db10', 4, (1, 1, 40, 24)),
                                       ('db8', 6, (1, 1, 64, 33))]):
    mode = 'periodization'
    x = f32(rng.randn(*shape))
    xt = torch.tensor(x, requires_grad=True)
    xfm, ifm = pw.DWTForward(J=J, wave=wave, mode=mode), pw.DWTInverse(wave=wave, mode=mode)
    yl, yh = xfm(xt)
    rec = ifm((yl, yh))
    gl, gh = f32(rng.randn(*yl.shape)), [f32(rng.randn(*h.shape)) for h in yh]
    dx, = torch.autograd.grad((yl * torch.tensor(gl)).sum() + sum((h * torch.tensor(g)).sum() for h, g in zip(yh, gh)), xt)
    ylr = yl.detach().clone().requires_grad_(True)
    yhr = [h.detach().clone().requires_grad_(True) for h in yh]
    gy = f32(rng.randn(*rec.shape))
    grads = torch.autograd.grad((ifm((ylr, yhr)) * torch.tensor(gy)).sum(), [ylr] + yhr)
    h = [npy(getattr(xfm, n)).ravel() for n in ('h0_col', 'h1_col', 'h0_row', 'h1_row')]
    g = [npy(getattr(ifm, n)).ravel() for n in ('g0_col', 'g1_col', 'g0_row', 'g1_row')]
    oyl, oyh = wo.dwt_forward(x, J, *h, mode)
    assert rel(oyl, npy(yl)) < TOL and all(rel(a, npy(b)) < TOL for a, b in zip(oyh, yh)), (wave, 'fwd')
    assert rel(wo.dwt_inverse(npy(yl), [npy(t) for t in yh], *g, mode), npy(rec)) < TOL, (wave, 'inv')
    shapes = [x.shape[-2:]] + [t.shape[-2:] for t in yh[:-1]]
    d = gl
    for j in range(J - 1, -1, -1):
        d = wo.afb2d_level_backward(d, gh[j], *h, mode, shapes[j])
    assert rel(d, npy(dx)) < TOL, (wave, 'afb bwd', rel(d, npy(dx)))
    arrs = dict(x=x, yl=npy(yl), rec=npy(rec), gl=gl, gy=gy, dx=npy(dx), dyl=npy(grads[0]))
    for j in range(J):
        arrs['yh%d' % j] = npy(yh[j]); arrs['gh%d' % j] = gh[j]; arrs['dyh%d' % j] = npy(grads[1 + j])
    save('dwt_r3_%d' % ci, dict(kind='dwt', wave=wave, mode=mode, J=J, shape=list(shape)), **arrs)
    print('dwt', wave, J, shape, 'ok')

# ---- 1-D periodization shorter than the filter
for ci, (wave, J, shape) in enumerate([('db8', 1, (2, 2, 8)), ('db5', 3, (1, 2, 20)), ('db10', 2, (1, 1, 7))]):
    mode = 'periodization'
    x = rng.randn(*shape)
    xt = torch.tensor(x, requires_grad=True)
    xfm, ifm = pw.DWT1DForward(J=J, wave=wave, mode=mode), pw.DWT1DInverse(wave=wave, mode=mode)
    yl, yh = xfm(xt)
    rec = ifm((yl, yh))
    h0, h1 = npy(xfm.h0).ravel(), npy(xfm.h1).ravel()
    g0, g1 = npy(ifm.g0).ravel(), npy(ifm.g1).ravel()
    oyl, oyh = wo.dwt1d_forward(x, J, h0, h1, mode)
    assert rel(oyl, npy(yl)) < TOL and all(rel(a, npy(b)) < TOL for a, b in zip(oyh, yh)), (wave, mode)
    assert rel(wo.dwt1d_inverse(npy(yl), [npy(t) for t in yh], g0, g1, mode), npy(rec)) < TOL
    gl, gh = rng.randn(*yl.shape), [rng.randn(*t.shape) for t in yh]
    dx, = torch.autograd.grad((yl * torch.tensor(gl)).sum() + sum((a * torch.tensor(b)).sum() for a, b in zip(yh, gh)), xt)
    ylr = torch.tensor(npy(yl), requires_grad=True)
    yhr = [torch.tensor(npy(t), requires_grad=True) for t in yh]
    gy = rng.randn(*rec.shape)
    grads = torch.autograd.grad((ifm((ylr, yhr)) * torch.tensor(gy)).sum(), [ylr] + yhr)
    arrs = dict(x=x, yl=npy(yl), rec=npy(rec), gl=gl, dx=npy(dx), gy=gy, dyl=npy(grads[0]))
    for j in range(J):
        arrs['yh%d' % j] = npy(yh[j]); arrs['gh%d' % j] = gh[j]; arrs['dyh%d' % j] = npy(grads[1 + j])
    save('ext_dwt1d_r3_%d' % ci, dict(kind='dwt1d', wave=wave, mode=mode, J=J, shape=list(shape)), **arrs)
    print('dwt1d', wave, J, shape, 'ok')

# ---- function-level afb1d / sfb1d, periodization, odd tap counts and short signals (tensor taps = already reversed)
arrs, meta_cases = {}, []
for L in (3, 5, 7, 10, 14):
    for n, d in ((9, 2), (12, 3), (2, 3), (3, 2), (4, 3)):
        shape = [2, 2, 6, 6]
        shape[d] = n
        x = rng.randn(*shape)
        h0, h1 = rng.randn(L), rng.randn(L)
        r = rdl.afb1d(torch.tensor(x), torch.tensor(h0), torch.tensor(h1), mode='periodization', dim=d)
        lo, hi = wo.afb1d(x, h0, h1, 'periodization', axis=d)
        rr = npy(r).reshape(shape[0], shape[1], 2, *r.shape[2:])
        assert rel(lo, rr[:, :, 0]) < TOL and rel(hi, rr[:, :, 1]) < TOL, (L, n, d)
        key = 'L%d_n%d_d%d' % (L, n, d)
        arrs[key + '_x'], arrs[key + '_h0'], arrs[key + '_h1'], arrs[key + '_lohi'] = x, h0, h1, npy(r)
        c = dict(key=key, L=L, n=n, dim=d)
        if L % 2 == 0:
            g0, g1 = rng.randn(L), rng.randn(L)
            lo_t, hi_t = torch.tensor(rr[:, :, 0].copy()), torch.tensor(rr[:, :, 1].copy())
            y = rdl.sfb1d(lo_t, hi_t, torch.tensor(g0), torch.tensor(g1), mode='periodization', dim=d)
            assert rel(wo.sfb1d(rr[:, :, 0], rr[:, :, 1], g0, g1, 'periodization', axis=d), npy(y)) < TOL, (L, n, d)
            arrs[key + '_g0'], arrs[key + '_g1'], arrs[key + '_y'] = g0, g1, npy(y)
            c['syn'] = True
        meta_cases.append(c)
np.savez_compressed(os.path.join(GOLD, 'ext_afb1d_per.npz'), **arrs)   # float64: these are tiny
index['ext_afb1d_per'] = dict(kind='afb1d_per', cases=meta_cases)
print('afb1d periodization (odd L, short signals):', len(meta_cases), 'cases ok')

json.dump(index, open(idx_path, 'w'), indent=1, sort_keys=True)
print('index updated')
