"""DTCWT filter preparation and the 1-D filter primitives of the reference (pytorch_wavelets/dtcwt/lowlevel.py:58-295).

The transforms do not call these: every level is one fused kernel (csrc/wl_dtcwt_tile.h).  They exist because the
reference exposes and tests them on their own (tests/test_colfilter.py, test_coldfilt.py, ... upstream); each one is
one to four launches of the engine's single-axis correlation kernel (wl_corr1d)."""
import numpy as np
import torch

from .. import ops


def prep_filt(h, c, transpose=False):
    """Column-vector filter, REVERSED, shape (c,1,L,1) (or (c,1,1,L) with transpose), default dtype."""
    h = np.asarray(h.detach().cpu().numpy() if isinstance(h, torch.Tensor) else h, dtype=np.float64)
    h = h.reshape(-1)[::-1].reshape(1, 1, -1, 1)
    h = np.repeat(h, repeats=c, axis=0)
    if transpose:
        h = h.transpose((0, 1, 3, 2))
    return torch.tensor(np.copy(h), dtype=torch.get_default_dtype())


def _sym_or_zero(mode):
    return ops.EXT_SYM if mode == 'symmetric' else ops.EXT_ZERO


def _empty(X):
    return X is None or X.shape == torch.Size([])


def _placeholder(X, h):
    """What the reference returns for a missing input (dtcwt/lowlevel.py:71-72 etc.).  Upstream writes ``X.device`` there and
    so raises AttributeError for ``X is None``; the filter's device is used instead."""
    return torch.zeros(1, 1, 1, 1, device=h.device if X is None else X.device)


def colfilter(X, h, mode='symmetric'):
    """Filter the columns of X (along H) with the odd-length taps h, same size out (reference dtcwt/lowlevel.py:70-80):
    Y[i] = sum_j h[j] * ext(X, i + j - L//2), symmetric extension for mode 'symmetric', zero padding otherwise."""
    if _empty(X):
        return _placeholder(X, h)
    L = h.numel()
    m = L // 2
    return ops.corr1d(X, 2, h, None, X.shape[2] + 2 * m - L + 1, -m, 1, 1, _sym_or_zero(mode))


def rowfilter(X, h, mode='symmetric'):
    """Filter the rows of X (along W) (reference dtcwt/lowlevel.py:83-94)."""
    if _empty(X):
        return _placeholder(X, h)
    L = h.numel()
    m = L // 2
    return ops.corr1d(X, 3, h, None, X.shape[3] + 2 * m - L + 1, -m, 1, 1, _sym_or_zero(mode))


def _dfilt(X, ha, hb, highpass, dim, what):
    n = X.shape[dim]
    if n % 4 != 0:
        raise ValueError('No. of {} in X must be a multiple of 4\nX was {}'.format(what, X.shape))
    m = ha.numel()
    shape = list(X.shape)
    shape[dim] = n // 2
    out = torch.empty(shape, dtype=X.dtype, device=X.device)
    # Ya[k] = sum_t ha[t] sym(X, 4k+2t+2-m) and Yb[k] = sum_t hb[t] sym(X, 4k+2t+3-m), interleaved (swapped for highpass)
    ops.corr1d(X, dim, ha, None, n // 4, 2 - m, 4, 2, ops.EXT_SYM, out=(out, None), out_offset=1 if highpass else 0,
               out_stride=2, ny=n // 2)
    ops.corr1d(X, dim, hb, None, n // 4, 3 - m, 4, 2, ops.EXT_SYM, out=(out, None), out_offset=0 if highpass else 1,
               out_stride=2, ny=n // 2)
    return out


def coldfilt(X, ha, hb, highpass=False, mode='symmetric'):
    """Dual-tree decimating column filter (reference dtcwt/lowlevel.py:97-122); rows must be a multiple of 4."""
    if _empty(X):
        return _placeholder(X, ha)
    if mode != 'symmetric':
        raise NotImplementedError()
    return _dfilt(X, ha, hb, highpass, 2, 'rows')


def rowdfilt(X, ha, hb, highpass=False, mode='symmetric'):
    """Dual-tree decimating row filter (reference dtcwt/lowlevel.py:125-151); columns must be a multiple of 4."""
    if _empty(X):
        return _placeholder(X, ha)
    if mode != 'symmetric':
        raise NotImplementedError()
    return _dfilt(X, ha, hb, highpass, 3, 'cols')


def _ifilt(X, ha, hb, highpass, dim, what):
    n = X.shape[dim]
    if n % 2 != 0:
        raise ValueError('No. of {} in X must be a multiple of 2.\nX was {}'.format(what, X.shape))
    m2 = ha.numel() // 2
    even, odd = (0, 2, m2), (1, 2, m2)           # hae = ha[0::2], hao = ha[1::2]
    if m2 % 2 == 0:
        f = ((ha, even), (hb, even), (ha, odd), (hb, odd))
        o = (1, 0, 3, 2) if highpass else (0, 1, 2, 3)
    else:
        f = ((ha, odd), (hb, odd), (ha, even), (hb, even))
        o = (2, 1, 2, 1) if highpass else (1, 2, 1, 2)
    shape = list(X.shape)
    shape[dim] = 2 * n
    out = torch.empty(shape, dtype=X.dtype, device=X.device)
    # Y[4q+s] = sum_{t<m2} f_s[t] * sym(X, o_s - m2 + 2(q+t))
    for s in range(4):
        ops.corr1d(X, dim, f[s][0], None, n // 2, o[s] - m2, 2, 2, ops.EXT_SYM, taps=f[s][1], out=(out, None),
                   out_offset=s, out_stride=4, ny=2 * n)
    return out


def colifilt(X, ha, hb, highpass=False, mode='symmetric'):
    """Dual-tree interpolating column filter (reference dtcwt/lowlevel.py:154-195); rows must be even."""
    if _empty(X):
        return _placeholder(X, ha)
    return _ifilt(X, ha, hb, highpass, 2, 'rows')


def rowifilt(X, ha, hb, highpass=False, mode='symmetric'):
    """Dual-tree interpolating row filter (reference dtcwt/lowlevel.py:198-239); columns must be even."""
    if _empty(X):
        return _placeholder(X, ha)
    return _ifilt(X, ha, hb, highpass, 3, 'cols')


def q2c(y, dim=-1):
    """Quads -> two complex sub-images ((z1r, z1i), (z2r, z2i)) (reference dtcwt/lowlevel.py:243-260): index shuffles
    and three adds on a quarter-size tensor - left to the tensor library, as upstream does."""
    y = y / np.sqrt(2)
    a, b = y[:, :, 0::2, 0::2], y[:, :, 0::2, 1::2]
    c, d = y[:, :, 1::2, 0::2], y[:, :, 1::2, 1::2]
    return ((a - d, b + c), (a + d, b - c))


def c2q(w1, w2):
    """Two complex sub-images -> quads (reference dtcwt/lowlevel.py:263-295)."""
    w1r, w1i = w1
    w2r, w2i = w2
    b, ch, r, c = w1r.shape
    y = w1r.new_zeros((b, ch, r * 2, c * 2))
    y[:, :, ::2, ::2] = w1r + w2r
    y[:, :, ::2, 1::2] = w1i + w2i
    y[:, :, 1::2, ::2] = w1i - w2i
    y[:, :, 1::2, 1::2] = -w1r + w2r
    return y / np.sqrt(2)
