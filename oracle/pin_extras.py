"""Pin the oracle's restatement of the 1-D DWT, the a-trous bank and the DTCWT 1-D primitives against the REAL reference
and write tests/golden/ext_*.npz (merged into index.json; the other fixtures are left alone).

    PYTHONPATH=tools/ref_shim:/root/reference python oracle/pin_extras.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import wavelet_oracle as wo   # noqa: E402
import pytorch_wavelets as pw             # noqa: E402  (the reference)
from pytorch_wavelets.dwt import lowlevel as rdl           # noqa: E402
from pytorch_wavelets.dwt.transform2d import SWTForward     # noqa: E402
from pytorch_wavelets.dtcwt import lowlevel as rtl          # noqa: E402
from pytorch_wavelets.dtcwt.coeffs import biort as _biort, qshift as _qshift   # noqa: E402

torch.set_default_dtype(torch.float64)
GOLD = os.path.join(ROOT, 'tests', 'golden')
idx_path = os.path.join(GOLD, 'index.json')
index = json.load(open(idx_path))
rng = np.random.RandomState(77)
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


# ---- 1-D DWT (dwt/transform1d.py) incl. the custom backward passes
cases = [('db4', 'symmetric', 3, (2, 3, 101)), ('db2', 'zero', 2, (1, 2, 64)), ('db3', 'periodization', 2, (2, 1, 50)),
         ('haar', 'reflect', 2, (1, 2, 33)), ('sym5', 'periodization', 3, (1, 1, 97))]
for ci, (wave, mode, J, shape) in enumerate(cases):
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
    save('ext_dwt1d_%d' % ci, dict(kind='dwt1d', wave=wave, mode=mode, J=J, shape=list(shape)), **arrs)
    print('dwt1d', wave, mode, J, shape, 'ok')

# ---- stationary transform, one level (the only configuration upstream's SWTForward completes) + dilated levels of the bank
for ci, (wave, mode, shape) in enumerate([('db2', 'periodic', (1, 2, 16, 24)), ('db4', 'symmetric', (2, 1, 33, 30)),
                                          ('haar', 'zero', (1, 2, 8, 9)), ('db3', 'reflect', (1, 1, 40, 28))]):
    x = rng.randn(*shape)
    m = SWTForward(J=1, wave=wave, mode=mode)
    y = m(torch.tensor(x))[0]
    hb = [npy(getattr(m, n)).ravel() for n in ('h0_col', 'h1_col', 'h0_row', 'h1_row')]
    assert rel(wo.afb2d_atrous(x, *hb, mode, 1), npy(y)) < TOL
    filts = (m.h0_col, m.h1_col, m.h0_row, m.h1_row)
    y2 = rdl.afb2d_atrous(torch.tensor(x), filts, mode, 2)          # the level-2 operator itself works upstream
    assert rel(wo.afb2d_atrous(x, *hb, mode, 2), npy(y2)) < TOL
    save('ext_swt_%d' % ci, dict(kind='swt', wave=wave, mode=mode, shape=list(shape)), x=x, y=npy(y), y_dil2=npy(y2))
    print('swt', wave, mode, shape, 'ok')

# ---- DTCWT primitives (dtcwt/lowlevel.py:70-295)
h0o, g0o, h1o, g1o = _biort('near_sym_b')
h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = _qshift('qshift_b')
P = rtl.prep_filt
X = rng.randn(2, 3, 16, 24)
Xt = torch.tensor(X)
out = dict(X=X)
res = {
    'colfilter': rtl.colfilter(Xt, P(h1o, 1)), 'rowfilter': rtl.rowfilter(Xt, P(h0o, 1)),
    'colfilter_zero': rtl.colfilter(Xt, P(h0o, 1), 'zero'),
    'coldfilt': rtl.coldfilt(Xt, P(h0b, 1), P(h0a, 1)), 'coldfilt_hp': rtl.coldfilt(Xt, P(h1b, 1), P(h1a, 1), True),
    'rowdfilt': rtl.rowdfilt(Xt, P(h0b, 1), P(h0a, 1)), 'rowdfilt_hp': rtl.rowdfilt(Xt, P(h1b, 1), P(h1a, 1), True),
    'colifilt': rtl.colifilt(Xt, P(g0b, 1), P(g0a, 1)), 'colifilt_hp': rtl.colifilt(Xt, P(g1b, 1), P(g1a, 1), True),
    'rowifilt': rtl.rowifilt(Xt, P(g0b, 1), P(g0a, 1)), 'rowifilt_hp': rtl.rowifilt(Xt, P(g1b, 1), P(g1a, 1), True),
}
r = lambda v: npy(P(v, 1)).ravel()   # noqa: E731
assert rel(wo.colfilter(X, r(h1o)), npy(res['colfilter'])) < TOL
assert rel(wo.rowfilter(X, r(h0o)), npy(res['rowfilter'])) < TOL
assert rel(wo.colfilter(X, r(h0o), 'zero'), npy(res['colfilter_zero'])) < TOL
assert rel(wo.coldfilt(X, r(h0b), r(h0a)), npy(res['coldfilt'])) < TOL
assert rel(wo.coldfilt(X, r(h1b), r(h1a), True), npy(res['coldfilt_hp'])) < TOL
assert rel(wo.rowdfilt(X, r(h0b), r(h0a)), npy(res['rowdfilt'])) < TOL
assert rel(wo.colifilt(X, r(g0b), r(g0a)), npy(res['colifilt'])) < TOL
assert rel(wo.colifilt(X, r(g1b), r(g1a), True), npy(res['colifilt_hp'])) < TOL
assert rel(wo.rowifilt(X, r(g1b), r(g1a), True), npy(res['rowifilt_hp'])) < TOL
(a, b), (c, d) = rtl.q2c(Xt)
out.update({k: npy(v) for k, v in res.items()})
out.update(q2c_1r=npy(a), q2c_1i=npy(b), q2c_2r=npy(c), q2c_2i=npy(d), c2q=npy(rtl.c2q((a, b), (c, d))))
save('ext_prims', dict(kind='prims', biort='near_sym_b', qshift='qshift_b'), **out)
print('primitives ok')

# ---- function-level afb1d / sfb1d (interleaved channels, dwt/lowlevel.py:91-172, :226-271)
x = rng.randn(1, 2, 12, 21).astype(np.float32)   # list filters become float32 taps upstream
w = __import__('pywt').Wavelet('db3')
lohi = rdl.afb1d(torch.tensor(x), w.dec_lo, w.dec_hi, mode='symmetric', dim=3)
lo_, hi_ = lohi[:, ::2].contiguous(), lohi[:, 1::2].contiguous()
y = rdl.sfb1d(lo_, hi_, w.rec_lo, w.rec_hi, mode='symmetric', dim=3)
save('ext_afb1d', dict(kind='afb1d', wave='db3', mode='symmetric'), x=x, lohi=npy(lohi), y=npy(y))
# ---- ScatLayerj2 (scatternet/layers.py:82-172) incl. its hand-written backward pass
for ci, (shape, comb) in enumerate([((2, 3, 32, 40), False), ((1, 3, 32, 32), True), ((1, 2, 35, 29), False),
                                    ((1, 1, 64, 48), False)]):
    x = rng.randn(*shape)
    m = pw.ScatLayerj2(combine_colour=comb)
    xt = torch.tensor(x, requires_grad=True)
    Z = m(xt)
    f = [npy(getattr(m, n)).ravel() for n in ('h0o', 'h1o', 'h0a', 'h0b', 'h1a', 'h1b')]
    assert rel(wo.scat_layer_j2_forward(x, *f, magbias=m.magbias, combine_colour=comb), npy(Z)) < TOL
    gz = rng.randn(*Z.shape)
    dx, = torch.autograd.grad((Z * torch.tensor(gz)).sum(), xt)
    save('ext_scatj2_%d' % ci, dict(kind='scatj2', shape=list(shape), combine_colour=comb), x=x, Z=npy(Z), gz=gz, dx=npy(dx))
    print('scatj2', shape, comb, 'ok')

# ---- rotationally symmetric variants (near_sym_b_bp / qshift_b_bp): ScatLayer and ScatLayerj2, forward + backward.
# (No separate numpy restatement: the engine's rot path is the reference's own decomposition into the single-axis
#  primitives pinned above; these goldens pin the composition.)
rot_cases = [('ScatLayer', dict(biort='near_sym_b_bp'), (2, 3, 32, 40)),
             ('ScatLayer', dict(biort='near_sym_b_bp', combine_colour=True), (1, 3, 33, 31)),
             ('ScatLayer', dict(biort='near_sym_b_bp', mode='zero'), (1, 2, 24, 20)),
             ('ScatLayerj2', dict(biort='near_sym_b_bp', qshift='qshift_b_bp'), (1, 2, 32, 40)),
             ('ScatLayerj2', dict(biort='near_sym_b_bp', qshift='qshift_b_bp', combine_colour=True), (1, 3, 35, 29))]
for ci, (cls, kw, shape) in enumerate(rot_cases):
    x = rng.randn(*shape)
    m = getattr(pw, cls)(**kw)
    xt = torch.tensor(x, requires_grad=True)
    Z = m(xt)
    gz = rng.randn(*Z.shape)
    dx, = torch.autograd.grad((Z * torch.tensor(gz)).sum(), xt)
    save('ext_rot_%d' % ci, dict(kind='rot', cls=cls, kwargs=kw, shape=list(shape)), x=x, Z=npy(Z), gz=gz, dx=npy(dx))
    print('rot', cls, kw, shape, 'ok')

json.dump(index, open(idx_path, 'w'), indent=1, sort_keys=True)
print('index updated:', sorted(k for k in index if k.startswith('ext_')))
