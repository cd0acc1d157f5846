"""Pin the oracle against the REAL reference and (re)generate tests/golden/*.npz.

Runs only where /root/reference is mounted (the authoring container):

    PYTHONPATH=tools/ref_shim:/root/reference python oracle/pin_against_reference.py

For every case below the reference module (fbcotter/pytorch_wavelets v1.3.0, run on CPU in
float64) is evaluated on a seeded input, ``oracle/wavelet_oracle.py`` must reproduce it to 1e-10
relative, and inputs + reference outputs are stored as small fixtures.  The fixtures are what
travels: the GPU box has no /root/reference.  ``tests/test_oracle_golden.py`` re-checks the
oracle against them; the ``-m gpu`` parity tests check the HIP path against them.
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

import pytorch_wavelets as pw            # noqa: E402  (the reference)
from pytorch_wavelets.dwt import lowlevel as ref_ll                 # noqa: E402
from pytorch_wavelets.scatternet import ScatLayer as RefScat        # noqa: E402

torch.set_default_dtype(torch.float64)
GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)
TOL = 1e-10
index = {}


def rel(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(b).max()
    return float(np.abs(a - b).max() / (d if d > 0 else 1.0))


def npy(t):
    return t.detach().numpy()


def buf(m, name):
    return npy(getattr(m, name)).ravel()


BIG = 100000          # tensors above this many elements are stored as samples, not in full
NSAMP = 8192


def save(name, meta, **arrs):
    """Small cases are stored in full.  For the full-plane-size cases (512x512 / 256x256) the
    input is kept in full (float32) but every large output is stored as: 8192 seeded random flat
    positions + their values + the tensor's l2 norm and sum (keys ``k__idx/__val/__stat``), so
    the fixtures stay small; tests/_golden.py knows both forms."""
    out = {}
    srng = np.random.RandomState(99)
    for k, v in arrs.items():
        v = np.asarray(v)
        if v.size > BIG and k != 'x':
            flat = v.ravel()
            idx = np.sort(srng.choice(flat.size, NSAMP, replace=False)).astype(np.int64)
            out[k + '__idx'] = idx
            out[k + '__val'] = flat[idx]
            out[k + '__stat'] = np.array([np.sqrt((flat.astype(np.float64) ** 2).sum()),
                                          flat.astype(np.float64).sum()] + list(v.shape), dtype=np.float64)
        else:
            # reference ran in float64 and the oracle matched it to 1e-10 above; the stored copy is
            # rounded to float32 (6e-8 relative) to keep the fixtures small - the parity gate is 1e-5
            out[k] = v.astype(np.float32) if v.dtype == np.float64 else v
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    index[name] = meta


# ------------------------------------------------------------------------------- DWT
dwt_cases = [
    # (wave, mode, J, shape)
    ('haar', 'zero', 1, (1, 3, 64, 64)),              # BASELINE configs[0]
    ('db4', 'symmetric', 3, (2, 3, 64, 64)),          # configs[1] geometry, reduced
    ('db4', 'symmetric', 3, (1, 1, 512, 512)),        # configs[1] full plane size
    ('db8', 'periodization', 4, (1, 1, 128, 128)),    # configs[4] geometry, reduced
    ('db1', 'zero', 2, (2, 2, 32, 48)),
    ('db2', 'symmetric', 2, (2, 2, 37, 41)),
    ('db3', 'reflect', 2, (1, 2, 40, 33)),
    ('db3', 'periodization', 3, (2, 1, 63, 50)),
    ('db4', 'periodic', 2, (1, 2, 45, 52)),
    ('db4', 'zero', 3, (1, 2, 99, 100)),
    ('bior2.4', 'symmetric', 2, (1, 2, 50, 51)),
    ('bior3.1', 'periodization', 2, (1, 1, 34, 36)),
    ('sym6', 'reflect', 2, (1, 1, 64, 70)),
    ('coif2', 'periodic', 2, (1, 1, 57, 64)),
    ('db4', 'symmetric', 3, (1, 1, 127, 127)),
    ('db4', 'periodization', 3, (1, 1, 127, 126)),
    ('db12', 'symmetric', 2, (1, 1, 96, 80)),
    ('db8', 'periodization', 4, (1, 1, 64, 64)),      # Ne < L at the last level (fold quirk)
]
rng = np.random.RandomState(1234)
for ci, (wave, mode, J, shape) in enumerate(dwt_cases):
    x = rng.randn(*shape).astype(np.float32).astype(np.float64)
    xt = torch.tensor(x, requires_grad=True)
    xfm = pw.DWTForward(J=J, wave=wave, mode=mode)
    ifm = pw.DWTInverse(wave=wave, mode=mode)
    yl, yh = xfm(xt)
    rec = ifm((yl, yh))
    # reference custom backward of the forward transform
    gl = rng.randn(*yl.shape).astype(np.float32).astype(np.float64)
    gh = [rng.randn(*h.shape).astype(np.float32).astype(np.float64) for h in yh]
    loss = (yl * torch.tensor(gl)).sum() + sum((h * torch.tensor(g)).sum() for h, g in zip(yh, gh))
    dx, = torch.autograd.grad(loss, xt)
    # reference custom backward of the inverse transform
    ylr = yl.detach().clone().requires_grad_(True)
    yhr = [h.detach().clone().requires_grad_(True) for h in yh]
    rec2 = ifm((ylr, yhr))
    gy = rng.randn(*rec2.shape).astype(np.float32).astype(np.float64)
    grads = torch.autograd.grad((rec2 * torch.tensor(gy)).sum(), [ylr] + yhr)

    # oracle forward / inverse
    h = [buf(xfm, n) for n in ('h0_col', 'h1_col', 'h0_row', 'h1_row')]
    g = [buf(ifm, n) for n in ('g0_col', 'g1_col', 'g0_row', 'g1_row')]
    oyl, oyh = wo.dwt_forward(x, J, *h, mode)
    assert rel(oyl, npy(yl)) < TOL, (wave, mode, rel(oyl, npy(yl)))
    for a, b in zip(oyh, yh):
        assert rel(a, npy(b)) < TOL, (wave, mode)
    orec = wo.dwt_inverse(npy(yl), [npy(t) for t in yh], *g, mode)
    assert rel(orec, npy(rec)) < TOL, (wave, mode, rel(orec, npy(rec)))
    # oracle backward of forward: chain AFB2D backward through the levels
    shapes = [x.shape[-2:]] + [t.shape[-2:] for t in yh[:-1]]
    d = gl
    for j in range(J - 1, -1, -1):
        d = wo.afb2d_level_backward(d, gh[j], *h, mode, shapes[j])
    assert rel(d, npy(dx)) < TOL, (wave, mode, 'afb bwd', rel(d, npy(dx)))
    # oracle backward of inverse: chain SFB2D backward from the finest level down
    d = gy
    ograds_h = []
    ll_shapes = [t.shape[-2:] for t in yh]
    for j in range(J):
        dlow, dhigh = wo.sfb2d_level_backward(d, *g, mode)
        ograds_h.append(dhigh)
        d = dlow
        if j + 1 < J:
            # the crop (ll[..., :-1, :]) before the finer level pads the gradient with zeros
            Lg = g[0].size
            tgt = [2 * v if mode == 'periodization' else 2 * v - Lg + 2 for v in ll_shapes[j + 1]]
            if d.shape[-2] < tgt[0] or d.shape[-1] < tgt[1]:
                d = np.pad(d, ((0, 0), (0, 0), (0, tgt[0] - d.shape[-2]), (0, tgt[1] - d.shape[-1])))
    assert rel(d, npy(grads[0])) < TOL, (wave, mode, 'sfb bwd low', rel(d, npy(grads[0])))
    for a, b in zip(ograds_h, grads[1:]):
        assert rel(a, npy(b)) < TOL, (wave, mode, 'sfb bwd high')
    big = x.size > BIG
    arrs = dict(x=x.astype(np.float32), yl=npy(yl), rec=npy(rec))
    if not big:
        arrs.update(gl=gl.astype(np.float32), gy=gy.astype(np.float32), dx=npy(dx), dyl=npy(grads[0]))
    for j in range(J):
        arrs['yh%d' % j] = npy(yh[j])
        if not big:
            arrs['gh%d' % j] = gh[j].astype(np.float32)
            arrs['dyh%d' % j] = npy(grads[1 + j])
    save('dwt_%02d' % ci, dict(kind='dwt', wave=wave, mode=mode, J=J, shape=list(shape)), **arrs)
    print('dwt', wave, mode, J, shape, 'ok')

# custom separate row/col filters (quirk Q1) + None highs in the inverse
w1, w2 = pw.DWTForward(wave='db1'), pw.DWTForward(wave='db3')
import pywt as _shim                                                       # noqa: E402
wv = (_shim.Wavelet('db1').dec_lo, _shim.Wavelet('db1').dec_hi,
      _shim.Wavelet('db3').dec_lo, _shim.Wavelet('db3').dec_hi)
wvi = (_shim.Wavelet('db1').rec_lo, _shim.Wavelet('db1').rec_hi,
       _shim.Wavelet('db3').rec_lo, _shim.Wavelet('db3').rec_hi)
x = rng.randn(1, 2, 32, 32).astype(np.float32).astype(np.float64)
xfm = pw.DWTForward(J=2, wave=wv, mode='symmetric')
ifm = pw.DWTInverse(wave=wvi, mode='symmetric')
yl, yh = xfm(torch.tensor(x))
rec_none = ifm((yl, [None, yh[1]]))
h = [buf(xfm, n) for n in ('h0_col', 'h1_col', 'h0_row', 'h1_row')]
g = [buf(ifm, n) for n in ('g0_col', 'g1_col', 'g0_row', 'g1_row')]
oyl, oyh = wo.dwt_forward(x, 2, *h, 'symmetric')
assert rel(oyl, npy(yl)) < TOL and oyl.shape[-2] != oyl.shape[-1], oyl.shape
orec = wo.dwt_inverse(npy(yl), [None, npy(yh[1])], *g, 'symmetric')
assert rel(orec, npy(rec_none)) < TOL
save('dwt_q1', dict(kind='dwt_q1', mode='symmetric', J=2), x=x.astype(np.float32), yl=npy(yl),
     yh0=npy(yh[0]), yh1=npy(yh[1]), rec_none=npy(rec_none),
     **{'h%d' % i: np.asarray(v) for i, v in enumerate(wv)},
     **{'g%d' % i: np.asarray(v) for i, v in enumerate(wvi)})
print('dwt q1 ok')

# ------------------------------------------------------------------------------- DTCWT
dtcwt_cases = [
    # biort, qshift, J, shape, skip_hps, include_scale, mode
    ('near_sym_a', 'qshift_a', 3, (2, 3, 64, 64), False, False, 'symmetric'),   # configs[2] reduced
    ('near_sym_a', 'qshift_a', 3, (1, 1, 512, 512), False, False, 'symmetric'),  # full plane
    ('near_sym_b', 'qshift_b', 3, (1, 2, 48, 40), False, False, 'symmetric'),
    ('antonini', 'qshift_c', 3, (1, 1, 99, 100), False, False, 'symmetric'),
    ('legall', 'qshift_d', 2, (1, 2, 52, 37), False, False, 'symmetric'),
    ('near_sym_a', 'qshift_06', 4, (1, 1, 126, 126), False, False, 'symmetric'),
    ('near_sym_a', 'qshift_a', 1, (1, 2, 33, 31), False, False, 'symmetric'),
    ('near_sym_a', 'qshift_a', 3, (1, 2, 60, 44), [False, True, False], [True, False, True], 'symmetric'),
    ('near_sym_b', 'qshift_a', 2, (1, 1, 40, 40), False, False, 'zero'),
]
for ci, (biort, qshift, J, shape, skip, inc, mode) in enumerate(dtcwt_cases):
    x = (100 * rng.randn(*shape)).astype(np.float32).astype(np.float64)
    xt = torch.tensor(x, requires_grad=True)
    xfm = pw.DTCWTForward(biort=biort, qshift=qshift, J=J, skip_hps=skip, include_scale=inc, mode=mode)
    ifm = pw.DTCWTInverse(biort=biort, qshift=qshift, mode=mode)
    yl, yh = xfm(xt)
    hb = [buf(xfm, n) for n in ('h0o', 'h1o', 'h0a', 'h0b', 'h1a', 'h1b')]
    gb = [buf(ifm, n) for n in ('g0o', 'g1o', 'g0a', 'g0b', 'g1a', 'g1b')]
    oyl, oyh = wo.dtcwt_forward(x, J, *hb, skip_hps=skip, include_scale=inc, mode=mode)
    arrs = dict(x=x.astype(np.float32))
    if isinstance(yl, (list, tuple)):
        for j, (a, b) in enumerate(zip(oyl, yl)):
            if b.shape == torch.Size([]):
                assert a is None
            else:
                assert rel(a, npy(b)) < TOL, (biort, 'scale', j)
                arrs['scale%d' % j] = npy(b)
        low = [t for t in yl if t.shape != torch.Size([])][-1]
    else:
        assert rel(oyl, npy(yl)) < TOL, (biort, qshift, rel(oyl, npy(yl)))
        low = yl
    arrs['yl'] = npy(low)
    for j, (a, b) in enumerate(zip(oyh, yh)):
        if b.shape == torch.Size([]):
            assert a is None
        else:
            assert rel(a, npy(b)) < TOL, (biort, qshift, 'yh', j, rel(a, npy(b)))
            arrs['yh%d' % j] = npy(b)
    rec = ifm((low, yh))
    orec = wo.dtcwt_inverse(npy(low), [None if t.shape == torch.Size([]) else npy(t) for t in yh],
                            *gb, mode=mode)
    assert rel(orec, npy(rec)) < TOL, (biort, qshift, 'inv', rel(orec, npy(rec)))
    arrs['rec'] = npy(rec)
    # backward of forward (all outputs weighted by random cotangents) and of inverse
    outs = [low] + [t for t in yh if t.shape != torch.Size([])]
    cots = [rng.randn(*t.shape).astype(np.float32).astype(np.float64) for t in outs]
    dx, = torch.autograd.grad(sum((t * torch.tensor(c)).sum() for t, c in zip(outs, cots)), xt)
    big = x.size > BIG
    if not big:
        arrs['dx'] = npy(dx)
        for i, c in enumerate(cots):
            arrs['cot%d' % i] = c.astype(np.float32)
    lowr = low.detach().clone().requires_grad_(True)
    yhr = [t.detach().clone().requires_grad_(t.shape != torch.Size([])) for t in yh]
    rec2 = ifm((lowr, yhr))
    gy = rng.randn(*rec2.shape).astype(np.float32).astype(np.float64)
    gin = [lowr] + [t for t in yhr if t.requires_grad]
    try:
        gr = torch.autograd.grad((rec2 * torch.tensor(gy)).sum(), gin)
        if not big:
            arrs['gy'] = gy.astype(np.float32)
            for i, t in enumerate(gr):
                arrs['dinv%d' % i] = npy(t)
    except RuntimeError as e:
        # upstream bug: INV_J1.backward does not undo the 1-px crop done inside inv_j1
        # (transform_funcs.py:171-176) so autograd rejects the gradient shape
        print('   reference inverse-backward unavailable for this case:', str(e)[:60])
    save('dtcwt_%02d' % ci, dict(kind='dtcwt', biort=biort, qshift=qshift, J=J, shape=list(shape),
                                 skip_hps=skip, include_scale=inc, mode=mode), **arrs)
    print('dtcwt', biort, qshift, J, shape, 'ok')

# inverse with None entries / missing lowpass (tests/test_dtcwt.py:258-294 upstream)
x = (100 * rng.randn(1, 2, 64, 48)).astype(np.float32).astype(np.float64)
xfm = pw.DTCWTForward(J=3)
ifm = pw.DTCWTInverse()
yl, yh = xfm(torch.tensor(x))
gb = [buf(ifm, n) for n in ('g0o', 'g1o', 'g0a', 'g0b', 'g1a', 'g1b')]
rec_a = ifm((yl, [None, yh[1], yh[2]]))
rec_b = ifm((yl, [yh[0], None, yh[2]]))
rec_c = ifm((torch.zeros_like(yl), [yh[0], yh[1], yh[2]]))
assert rel(wo.dtcwt_inverse(npy(yl), [None, npy(yh[1]), npy(yh[2])], *gb), npy(rec_a)) < TOL
assert rel(wo.dtcwt_inverse(npy(yl), [npy(yh[0]), None, npy(yh[2])], *gb), npy(rec_b)) < TOL
save('dtcwt_none', dict(kind='dtcwt_none', J=3), x=x.astype(np.float32), yl=npy(yl), yh0=npy(yh[0]),
     yh1=npy(yh[1]), yh2=npy(yh[2]), rec_a=npy(rec_a), rec_b=npy(rec_b), rec_c=npy(rec_c))
print('dtcwt none ok')

# ------------------------------------------------------------------------------- ScatLayer
scat_cases = [
    ('near_sym_a', 'symmetric', 1e-2, False, (2, 3, 32, 32)),     # configs[3] reduced
    ('near_sym_a', 'symmetric', 1e-2, False, (1, 3, 256, 256)),   # configs[3] plane size
    ('near_sym_b', 'symmetric', 1e-2, False, (1, 4, 31, 36)),
    ('near_sym_a', 'zero', 1e-1, False, (1, 2, 40, 24)),
    ('near_sym_a', 'symmetric', 1e-2, True, (2, 3, 32, 28)),
]
for ci, (biort, mode, magbias, cc, shape) in enumerate(scat_cases):
    x = rng.randn(*shape).astype(np.float32).astype(np.float64)
    xt = torch.tensor(x, requires_grad=True)
    sl = RefScat(biort=biort, mode=mode, magbias=magbias, combine_colour=cc)
    Z = sl(xt)
    gz = rng.randn(*Z.shape).astype(np.float32).astype(np.float64)
    dx, = torch.autograd.grad((Z * torch.tensor(gz)).sum(), xt)
    h0o, h1o = npy(sl.h0o).ravel(), npy(sl.h1o).ravel()
    oZ, saved = wo.scat_layer_forward(x, h0o, h1o, mode, magbias, cc, return_saved=True)
    assert rel(oZ, npy(Z)) < TOL, (biort, mode, rel(oZ, npy(Z)))
    odx = wo.scat_layer_backward(gz, saved, h0o, h1o, mode, cc)
    odx = odx[:, :, :shape[2], :shape[3]]
    # odd sizes: the module's edge replication folds the pad gradient into the last row/col
    r, c = shape[2:]
    full = wo.scat_layer_backward(gz, saved, h0o, h1o, mode, cc)
    if r % 2:
        full[:, :, r - 1] += full[:, :, r]
        full = full[:, :, :r]
    if c % 2:
        full[:, :, :, c - 1] += full[:, :, :, c]
        full = full[:, :, :, :c]
    assert rel(full, npy(dx)) < TOL, (biort, mode, 'bwd', rel(full, npy(dx)))
    extra = {} if x.size > BIG / 2 else dict(gz=gz.astype(np.float32), dx=npy(dx))
    save('scat_%02d' % ci, dict(kind='scat', biort=biort, mode=mode, magbias=magbias,
                                combine_colour=cc, shape=list(shape)),
         x=x.astype(np.float32), Z=npy(Z), **extra)
    print('scat', biort, mode, shape, 'ok')

with open(os.path.join(GOLD, 'index.json'), 'w') as f:
    json.dump(index, f, indent=1, sort_keys=True)
print('all pinned; fixtures in', GOLD)
