"""Dump the PyWavelets 1.1.1 discrete filter banks to JSON (data only, no code).

Run with the interpreter that has pywt (this image: /opt/conda/bin/python3.9):
    /opt/conda/bin/python3.9 tools/dump_pywt_filters.py

The reference obtains its DWT taps from ``pywt.Wavelet(name)``
(/root/reference/pytorch_wavelets/dwt/transform2d.py:22-26, :91-95); PyWavelets is not
installed in the build interpreter nor on the GPU box, so the taps travel as a table.
"""
import json
import os
import pywt

out = {}
for name in pywt.wavelist(kind='discrete'):
    w = pywt.Wavelet(name)
    out[name] = dict(dec_lo=list(map(float, w.dec_lo)), dec_hi=list(map(float, w.dec_hi)),
                     rec_lo=list(map(float, w.rec_lo)), rec_hi=list(map(float, w.rec_hi)))
# aliases pywt accepts
out['haar'] = out['haar'] if 'haar' in out else out['db1']
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch_wavelets_amd', 'data',
                   'pywt_filters.json')
with open(dst, 'w') as f:
    json.dump(dict(pywt_version=pywt.__version__, wavelets=out), f)
print('wrote', dst, len(out), 'wavelets')
