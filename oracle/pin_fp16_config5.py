"""Golden for BASELINE configs[4] at reduced size: DWTForward(J=4, 'db8', 'periodization') on float16 data.

    PYTHONPATH=tools/ref_shim:/root/reference python oracle/pin_fp16_config5.py

The REAL reference (fbcotter/pytorch_wavelets, CPU) is run in float32 on the float16-ROUNDED input (SURVEY.md 8(d):
the reference's own fp16 CPU path differs from its fp32 path by 2.5e-3, so fp32-on-rounded-input is the expected
value and ~2e-3 the tolerance).  The numpy oracle must reproduce it to 1e-6 (float32 reference); the fixture
tests/golden/dwt_h16.npz is added to tests/golden/index.json without touching the other fixtures.
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

GOLD = os.path.join(ROOT, 'tests', 'golden')
rng = np.random.RandomState(516)
J, wave, mode, shape = 4, 'db8', 'periodization', (1, 2, 256, 264)   # W % 8 == 0: the four-element staging path
x16 = rng.randn(*shape).astype(np.float16)
x = torch.tensor(x16.astype(np.float32))
xfm = pw.DWTForward(J=J, wave=wave, mode=mode)
ifm = pw.DWTInverse(wave=wave, mode=mode)
with torch.no_grad():
    yl, yh = xfm(x)
    # the inverse consumes the coefficients as the engine will see them: rounded to float16
    yl16 = yl.half().float()
    yh16 = [h.half().float() for h in yh]
    rec = ifm((yl16, yh16))
h = [getattr(xfm, n).numpy().ravel().astype(np.float64) for n in ('h0_col', 'h1_col', 'h0_row', 'h1_row')]
g = [getattr(ifm, n).numpy().ravel().astype(np.float64) for n in ('g0_col', 'g1_col', 'g0_row', 'g1_row')]
oyl, oyh = wo.dwt_forward(x.double().numpy(), J, *h, mode)


def rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / np.abs(b).max())


assert rel(yl.numpy(), oyl) < 1e-5, rel(yl.numpy(), oyl)
for a, b in zip(yh, oyh):
    assert rel(a.numpy(), b) < 1e-5
orec = wo.dwt_inverse(yl16.double().numpy(), [t.double().numpy() for t in yh16], *g, mode)
assert rel(rec.numpy(), orec) < 1e-5
arrs = dict(x=x16, yl=yl.numpy(), rec=rec.numpy())
for j in range(J):
    arrs['yh%d' % j] = yh[j].numpy()
np.savez_compressed(os.path.join(GOLD, 'dwt_h16.npz'), **arrs)
idx_path = os.path.join(GOLD, 'index.json')
index = json.load(open(idx_path))
index['dwt_h16'] = dict(kind='dwt_fp16', wave=wave, mode=mode, J=J, shape=list(shape))
json.dump(index, open(idx_path, 'w'), indent=1, sort_keys=True)
print('dwt_h16 written:', {k: v.shape for k, v in arrs.items()})
