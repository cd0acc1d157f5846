"""Consolidate the DTCWT biorthogonal / q-shift tap tables into one .npz (data only).

The reference ships these taps as one .npz per family under
/root/reference/pytorch_wavelets/dtcwt/data/ and reads them in dtcwt/coeffs.py:17-31.
They are numeric tables (Kingsbury's published filters), so they travel as data:
    python tools/dump_dtcwt_filters.py
writes pytorch_wavelets_amd/data/dtcwt_filters.npz with keys '<family>/<tap-name>'.
"""
import glob
import os
import numpy as np

src = '/root/reference/pytorch_wavelets/dtcwt/data'
out = {}
for f in sorted(glob.glob(os.path.join(src, '*.npz'))):
    fam = os.path.splitext(os.path.basename(f))[0]
    with np.load(f) as d:
        for k in d.files:
            if d[k].dtype.kind not in 'fiu' or d[k].size == 0:
                continue   # MATLAB header strings, not taps
            out['%s/%s' % (fam, k)] = np.asarray(d[k], dtype=np.float64).ravel()
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch_wavelets_amd', 'data',
                   'dtcwt_filters.npz')
np.savez_compressed(dst, **out)
print('wrote', dst)
for k, v in out.items():
    print(k, v.shape)
