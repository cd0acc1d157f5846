"""ctypes wrapper of oracle/dwt_port.c (C/OpenMP fp32 port of the oracle's DWT; baseline infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libwl_port.so')
_MODES = {'zero': 0, 'symmetric': 1, 'periodization': 2, 'per': 2, 'reflect': 4, 'periodic': 6}
_L = None


def _lib():
    global _L
    if _L is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'dwt_port.c')):
            subprocess.check_call(['make', '-s', '-C', _HERE])
        _L = ctypes.CDLL(_SO)
        fp, lg, it = ctypes.POINTER(ctypes.c_float), ctypes.c_long, ctypes.c_int
        for f in (_L.wl_port_dwt_forward, _L.wl_port_dwt_inverse):
            f.restype = lg
            f.argtypes = [fp, fp, lg, it, it, it, fp, fp, it, it, it]
    return _L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(x, J, h0, h1, mode, threads=0):
    """x (N,C,H,W) float32 -> packed coefficients (N*C, per) float32 (yh_0 .. yh_{J-1}, yl per plane)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    N, C, H, W = x.shape
    h0 = np.ascontiguousarray(h0, dtype=np.float32); h1 = np.ascontiguousarray(h1, dtype=np.float32)
    m = _MODES[mode]
    per = _lib().wl_port_dwt_forward(_p(x), None, N * C, H, W, J, _p(h0), _p(h1), h0.size, m, threads)
    out = np.empty((N * C, per), dtype=np.float32)
    _lib().wl_port_dwt_forward(_p(x), _p(out), N * C, H, W, J, _p(h0), _p(h1), h0.size, m, threads)
    return out


def inverse(coeffs, shape, J, g0, g1, mode, threads=0):
    N, C, H, W = shape
    g0 = np.ascontiguousarray(g0, dtype=np.float32); g1 = np.ascontiguousarray(g1, dtype=np.float32)
    y = np.empty((N, C, H, W), dtype=np.float32)
    _lib().wl_port_dwt_inverse(_p(coeffs), _p(y), N * C, H, W, J, _p(g0), _p(g1), g0.size, _MODES[mode], threads)
    return y


def unpack(coeffs, shape, J, L, mode):
    """Packed coefficients -> (yl, [yh_j]) arrays in the reference's layout."""
    N, C, H, W = shape
    yh, off, h, w = [], 0, H, W
    for _ in range(J):
        h = (h + 1) // 2 if mode in ('per', 'periodization') else (h + L - 1) // 2
        w = (w + 1) // 2 if mode in ('per', 'periodization') else (w + L - 1) // 2
        yh.append(coeffs[:, off:off + 3 * h * w].reshape(N, C, 3, h, w))
        off += 3 * h * w
    return coeffs[:, off:off + h * w].reshape(N, C, h, w), yh


def fwd_inv(x, J, h0, h1, g0, g1, mode, threads=0):
    c = forward(x, J, h0, h1, mode, threads)
    return inverse(c, x.shape, J, g0, g1, mode, threads)
