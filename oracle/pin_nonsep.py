"""Pin the oracle's restatement of the non-separable one-level banks (afb2d_nonsep / sfb2d_nonsep and their prep_filt
functions) against the REAL reference, incl. the gradients autograd gives upstream, and write tests/golden/ext_nonsep_*.npz.

    PYTHONPATH=tools/ref_shim:/root/reference python oracle/pin_nonsep.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import wavelet_oracle as wo                      # noqa: E402
from pytorch_wavelets.dwt import lowlevel as rdl             # noqa: E402  (the reference)
import pywt                                                  # noqa: E402  (tools/ref_shim)

torch.set_default_dtype(torch.float64)
GOLD = os.path.join(ROOT, 'tests', 'golden')
idx_path = os.path.join(GOLD, 'index.json')
index = json.load(open(idx_path))
rng = np.random.RandomState(4242)
TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


cases = [('db2', None, 'zero', (2, 2, 20, 24)), ('db3', None, 'symmetric', (1, 2, 21, 30)), ('db4', 'db2', 'reflect', (1, 1, 33, 26)),
         ('db3', None, 'periodization', (2, 1, 24, 32)), ('db2', 'db4', 'periodization', (1, 2, 25, 31)), ('haar', None, 'zero', (1, 3, 9, 12)),
         ('random', None, 'zero', (1, 2, 18, 23)), ('db3', None, 'periodic', (1, 1, 16, 20))]
for ci, (wcol, wrow, mode, shape) in enumerate(cases):
    x = rng.randn(*shape)
    if wcol == 'random':   # arbitrary point-spread functions (not outer products)
        fa = torch.tensor(rng.randn(4, 1, 5, 4))
        fs = torch.tensor(rng.randn(4, 1, 5, 4))
    else:
        wc = pywt.Wavelet(wcol)
        wr = pywt.Wavelet(wrow) if wrow else wc
        fa = rdl.prep_filt_afb2d_nonsep(wc.dec_lo, wc.dec_hi, wr.dec_lo, wr.dec_hi)
        fs = rdl.prep_filt_sfb2d_nonsep(wc.rec_lo, wc.rec_hi, wr.rec_lo, wr.rec_hi)
    arrs = {'x': x, 'fa': fa.numpy(), 'fs': fs.numpy()}
    meta = {'kind': 'nonsep', 'mode': mode, 'shape': list(shape), 'wave': [wcol, wrow]}
    if mode != 'periodic':   # upstream's afb2d_nonsep has no periodic branch
        xt = torch.tensor(x, requires_grad=True)
        y = rdl.afb2d_nonsep(xt, fa, mode)
        oy = wo.afb2d_nonsep(x, fa.numpy(), mode)
        assert rel(oy, y.detach().numpy()) < TOL, (ci, 'afb')
        gy = rng.randn(*y.shape)
        dx, = torch.autograd.grad((y * torch.tensor(gy)).sum(), xt)
        arrs.update(y=y.detach().numpy(), gy=gy, dx=dx.numpy())
        c = y.detach().reshape(shape[0], shape[1], 4, y.shape[-2], y.shape[-1])
    else:
        c = torch.tensor(rng.randn(shape[0], shape[1], 4, shape[2] // 2 + 2, shape[3] // 2 + 2))
    ct = c.clone().requires_grad_(True)
    rec = rdl.sfb2d_nonsep(ct, fs, mode)
    orec = wo.sfb2d_nonsep(c.numpy(), fs.numpy(), mode)
    assert rel(orec, rec.detach().numpy()) < TOL, (ci, 'sfb')
    gr = rng.randn(*rec.shape)
    dc, = torch.autograd.grad((rec * torch.tensor(gr)).sum(), ct)
    arrs.update(c=c.numpy(), rec=rec.detach().numpy(), gr=gr, dc=dc.numpy())
    if wcol not in ('random',) and mode != 'periodic':   # perfect reconstruction of the separable-equivalent banks
        H, W = shape[2], shape[3]
        assert rel(rec.detach().numpy()[..., :H, :W], x) < 1e-9, (ci, 'pr')
    name = 'ext_nonsep_%02d' % ci
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **{k: np.asarray(v).astype(np.float32) for k, v in arrs.items()})
    index[name] = meta
    print(name, mode, shape, 'ok')
json.dump(index, open(idx_path, 'w'), indent=1, sort_keys=True)
print('oracle == reference (non-separable banks) to', TOL)
