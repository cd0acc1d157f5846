"""Two-entry-point stand-in for PyWavelets, used ONLY to import the read-only reference
(/root/reference) in the authoring container when generating golden vectors.

The reference's hot path touches exactly ``pywt.Wavelet(name).{dec_lo,dec_hi,rec_lo,rec_hi}``
(dwt/transform2d.py:22-26,91-95) and ``pywt.dwt_coeff_len`` (dwt/lowlevel.py:153).
Taps come from the table dumped from real pywt 1.1.1 (tools/dump_pywt_filters.py).
Never imported by the product package or by tests.
"""
import json
import os

_T = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..',
                                 'pytorch_wavelets_amd', 'data', 'pywt_filters.json')))['wavelets']


class Wavelet(object):
    def __init__(self, name):
        t = _T[name]
        self.name = name
        self.dec_lo, self.dec_hi = t['dec_lo'], t['dec_hi']
        self.rec_lo, self.rec_hi = t['rec_lo'], t['rec_hi']


def dwt_coeff_len(data_len, filter_len, mode='symmetric'):
    if mode in ('per', 'periodization'):
        return (data_len + 1) // 2
    return (data_len + filter_len - 1) // 2
