"""Filter-bank tables (layer L0 of SURVEY.md §1): data only, no arithmetic on the hot path.

* DWT taps: the reference asks PyWavelets for them (``pywt.Wavelet(name)``,
  /root/reference/pytorch_wavelets/dwt/transform2d.py:22-26, :91-95).  PyWavelets is not a
  dependency here; the 106 discrete wavelets of pywt 1.1.1 are carried in
  ``data/pywt_filters.json`` (tools/dump_pywt_filters.py).
* DTCWT taps: Kingsbury's biorthogonal / q-shift tables, one consolidated
  ``data/dtcwt_filters.npz`` (tools/dump_dtcwt_filters.py), served through ``biort``/``qshift``/
  ``level1`` with the reference's return orders (dtcwt/coeffs.py:34-117).
"""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
_PYWT = None
_DTCWT = None


def _pywt_table():
    global _PYWT
    if _PYWT is None:
        with open(os.path.join(_DATA, 'pywt_filters.json')) as f:
            _PYWT = json.load(f)['wavelets']
    return _PYWT


def _dtcwt_table():
    global _DTCWT
    if _DTCWT is None:
        with np.load(os.path.join(_DATA, 'dtcwt_filters.npz')) as d:
            _DTCWT = {k: d[k] for k in d.files}
    return _DTCWT


class Wavelet(object):
    """Minimal stand-in for ``pywt.Wavelet``: ``dec_lo, dec_hi, rec_lo, rec_hi`` lists."""

    def __init__(self, name):
        t = _pywt_table()
        if name not in t:
            raise ValueError("Unknown wavelet name '{}', check wavelist() for the list of available "
                             "builtin wavelets.".format(name))
        self.name = name
        self.dec_lo, self.dec_hi = list(t[name]['dec_lo']), list(t[name]['dec_hi'])
        self.rec_lo, self.rec_hi = list(t[name]['rec_lo']), list(t[name]['rec_hi'])
        self.dec_len = self.rec_len = len(self.dec_lo)

    @property
    def filter_bank(self):
        return self.dec_lo, self.dec_hi, self.rec_lo, self.rec_hi


def wavelist():
    return sorted(_pywt_table())


def is_wavelet_like(w):
    return all(hasattr(w, a) for a in ('dec_lo', 'dec_hi', 'rec_lo', 'rec_hi'))


def dwt_analysis_taps(wave):
    """(h0, h1) as stored in the reference's ``h0_col`` … buffers: dec_* reversed
    (dwt/lowlevel.py:956-975)."""
    w = Wavelet(wave) if isinstance(wave, str) else wave
    return (np.array(w.dec_lo, dtype=np.float64)[::-1].copy(),
            np.array(w.dec_hi, dtype=np.float64)[::-1].copy())


def dwt_synthesis_taps(wave):
    """(g0, g1) as stored in ``g0_col`` …: rec_* unreversed (dwt/lowlevel.py:902-922)."""
    w = Wavelet(wave) if isinstance(wave, str) else wave
    return np.array(w.rec_lo, dtype=np.float64), np.array(w.rec_hi, dtype=np.float64)


def _load(family, names):
    t = _dtcwt_table()
    try:
        return tuple(t['%s/%s' % (family, k)] for k in names)
    except KeyError:
        if not any(k.startswith(family + '/') for k in t):
            raise IOError("No such wavelet family: '{}'".format(family))
        raise ValueError('Wavelet does not define ({0}) coefficients'.format(', '.join(names)))


def level1(name, compact=False):
    """dtcwt/coeffs.py:41-77."""
    if compact:
        if name == 'near_sym_b_bp':
            return _load(name, ('h0o', 'g0o', 'h1o', 'g1o', 'h2o', 'g2o'))
        return _load(name, ('h0o', 'g0o', 'h1o', 'g1o'))
    return _load(name, ('h0a', 'h0b', 'g0a', 'g0b', 'h1a', 'h1b', 'g1a', 'g1b'))


def biort(name):
    """dtcwt/coeffs.py:34-38: (h0o, g0o, h1o, g1o)."""
    return level1(name, compact=True)


def qshift(name):
    """dtcwt/coeffs.py:80-117: (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b)."""
    if name == 'qshift_b_bp':
        return _load(name, ('h0a', 'h0b', 'g0a', 'g0b', 'h1a', 'h1b', 'g1a', 'g1b', 'h2a', 'h2b',
                            'g2a', 'g2b'))
    return _load(name, ('h0a', 'h0b', 'g0a', 'g0b', 'h1a', 'h1b', 'g1a', 'g1b'))


def prep_dtcwt(h):
    """Buffer form of a DTCWT tap vector: reversed column (dtcwt/lowlevel.py:58-67)."""
    return np.asarray(h, dtype=np.float64).ravel()[::-1].copy()


def dtcwt_forward_taps(biort_name, qshift_name):
    """(h0o,h1o,h0a,h0b,h1a,h1b) in buffer form (dtcwt/transform2d.py:58-75)."""
    h0o, _, h1o, _ = biort(biort_name)[:4]
    h0a, h0b, _, _, h1a, h1b, _, _ = qshift(qshift_name)[:8]
    return tuple(prep_dtcwt(v) for v in (h0o, h1o, h0a, h0b, h1a, h1b))


def dtcwt_inverse_taps(biort_name, qshift_name):
    """(g0o,g1o,g0a,g0b,g1a,g1b) in buffer form (dtcwt/transform2d.py:174-191)."""
    _, g0o, _, g1o = biort(biort_name)[:4]
    _, _, g0a, g0b, _, _, g1a, g1b = qshift(qshift_name)[:8]
    return tuple(prep_dtcwt(v) for v in (g0o, g1o, g0a, g0b, g1a, g1b))
