"""CPU oracle for the separable 2-D wavelet filterbank hot path (TEST INFRASTRUCTURE ONLY).

This module is a from-scratch numpy restatement of the *algorithm* of
fbcotter/pytorch_wavelets for the path BASELINE.json names (DWTForward / DWTInverse /
DTCWTForward / DTCWTInverse / ScatLayer).  It is the checker the HIP kernels are compared with.

    * It is NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
      ``cpu_baseline`` leg of ``bench.py`` may import it.  ``pytorch_wavelets_amd`` never does.
    * Parity is PINNED: ``oracle/pin_against_reference.py`` runs the real reference
      (imported from /root/reference with the 2-function pywt stand-in under tools/ref_shim)
      on seeded inputs, asserts this module reproduces it, and writes the golden vectors under
      ``tests/golden/`` which ``tests/test_oracle_golden.py`` re-checks without the reference.

Every function states the closed form it evaluates and cites the reference lines (paths
relative to /root/reference/pytorch_wavelets/) whose behaviour it restates.  Arrays are
(N, C, H, W) like the reference; arithmetic is float64 unless the caller passes another dtype.

Filter-argument conventions (chosen to equal what the reference's *functions* receive):
    * DWT analysis taps ``h0, h1``: the arrays held in the module buffers ``h0_col`` … i.e. the
      pywt ``dec_lo/dec_hi`` lists REVERSED (dwt/lowlevel.py:956-975).  Synthesis taps
      ``g0, g1``: ``rec_lo/rec_hi`` as they are (dwt/lowlevel.py:902-922).
    * DTCWT taps: the buffer contents, i.e. the table vectors REVERSED (dtcwt/lowlevel.py:58-67).
"""
import numpy as np

MODE_INTS = {'zero': 0, 'symmetric': 1, 'per': 2, 'periodization': 2, 'constant': 3,
             'reflect': 4, 'replicate': 5, 'periodic': 6}
INT_MODES = {0: 'zero', 1: 'symmetric', 2: 'periodization', 3: 'constant', 4: 'reflect',
             5: 'replicate', 6: 'periodic'}


def mode_to_int(mode):
    """dwt/lowlevel.py:274-290 (same codes, same error text)."""
    if mode not in MODE_INTS:
        raise ValueError("Unkown pad type: {}".format(mode))
    return MODE_INTS[mode]


def int_to_mode(mode):
    """dwt/lowlevel.py:293-309."""
    if mode not in INT_MODES:
        raise ValueError("Unkown pad type: {}".format(mode))
    return INT_MODES[mode]


# ----------------------------------------------------------------------------------------------
# boundary extension
# ----------------------------------------------------------------------------------------------
def ext_index(idx, n, mode):
    """Map extended sample positions ``idx`` (any integers) of a length-``n`` signal to source
    positions.  Returns ``(src, valid)``; where ``valid`` is False the sample is 0.

    zero       : 0 outside [0,n)                              (F.pad, dwt/lowlevel.py:85-86)
    symmetric  : half-sample mirror, period 2n                (utils.py:146-163 with ±0.5 limits,
                                                               dwt/lowlevel.py:38-44; utils.py:166-174)
    reflect    : whole-sample mirror, period 2n-2             (F.pad 'reflect', dwt/lowlevel.py:83-84)
    periodic   : idx mod n                                    (np.pad 'wrap', dwt/lowlevel.py:62-72)
    """
    idx = np.asarray(idx, dtype=np.int64)
    if mode == 'zero':
        valid = (idx >= 0) & (idx < n)
        return np.clip(idx, 0, n - 1), valid
    valid = np.ones(idx.shape, dtype=bool)
    if mode == 'symmetric':
        i = np.mod(idx, 2 * n)
        return np.where(i < n, i, 2 * n - 1 - i), valid
    if mode == 'reflect':
        if n == 1:
            return np.zeros_like(idx), valid
        p = 2 * n - 2
        i = np.mod(idx, p)
        return np.where(i < n, i, p - i), valid
    if mode == 'periodic':
        return np.mod(idx, n), valid
    raise ValueError("Unkown pad type: {}".format(mode))


def _take_ext(x, idx, mode, axis):
    """x[..., ext(idx), ...] along ``axis`` with zero fill where invalid."""
    n = x.shape[axis]
    src, valid = ext_index(idx, n, mode)
    out = np.take(x, src, axis=axis)
    if not valid.all():
        shp = [1] * x.ndim
        shp[axis] = -1
        out = out * valid.reshape(shp).astype(x.dtype)
    return out


# ----------------------------------------------------------------------------------------------
# DWT 1-D banks
# ----------------------------------------------------------------------------------------------
def dwt_coeff_len(n, L, mode):
    """pywt.dwt_coeff_len as used at dwt/lowlevel.py:153 (periodization: ceil(n/2))."""
    if mode in ('per', 'periodization'):
        return (n + 1) // 2
    return (n + L - 1) // 2


def afb1d(x, h0, h1, mode='zero', axis=-1):
    """1-D analysis bank along ``axis``.  Returns (lo, hi).

    Restates dwt/lowlevel.py:91-172.  ``h0,h1`` are the stored (reversed) taps; with
    ``u = h[::-1]`` (the pywt dec_* order) the reference's strided cross-correlation is

      non-periodization (:151-170):  K=(N+L-1)//2, p=2(K-1)-N+L,
          y_b[k] = sum_m u_b[m] * ext_mode(x, 2k + (L-1-p//2) - m),     k in [0,K)
      periodization (:134-150):      xe = x (+ last sample repeated if N odd), Ne=len(xe),
          xr[i] = xe[(i+S) mod Ne], S = L//2 (roll by -L//2; the reference's roll(), :9-25, is two slices and
          degenerates into the identity once L//2 >= 2 Ne: S = 0 there),  xr0 = xr zero-extended,
          f[k]  = sum_m u_b[m] * xr0[2k-m],  k in [0, Ne/2+L//2)
          y_b[k] = f[k] + (f[k+Ne/2] if k < L//2 else 0),                k in [0,Ne/2)
    """
    if mode not in ('zero', 'symmetric', 'reflect', 'periodic', 'per', 'periodization'):
        raise ValueError("Unkown pad type: {}".format(mode))
    x = np.asarray(x)
    axis = axis % x.ndim
    u0 = np.asarray(h0, dtype=x.dtype).ravel()[::-1]
    u1 = np.asarray(h1, dtype=x.dtype).ravel()[::-1]
    L = u0.size
    N = x.shape[axis]
    outs = []
    if mode in ('per', 'periodization'):
        if N % 2 == 1:
            x = np.concatenate([x, np.take(x, [N - 1], axis=axis)], axis=axis)
            N += 1
        L2 = L // 2
        N2 = N // 2
        xr = np.take(x, np.mod(np.arange(N) + (L2 if L2 < 2 * N else 0), N), axis=axis)
        nfull = N2 + L2
        k = np.arange(nfull)
        for u in (u0, u1):
            f = 0
            for m in range(L):
                f = f + u[m] * _take_ext(xr, 2 * k - m, 'zero', axis)
            head = np.take(f, np.arange(N2), axis=axis).copy()
            nfold = min(L2, N2)
            sl_h = [slice(None)] * x.ndim
            sl_h[axis] = slice(0, nfold)
            sl_t = [slice(None)] * x.ndim
            sl_t[axis] = slice(N2, N2 + nfold)
            head[tuple(sl_h)] += f[tuple(sl_t)]
            outs.append(head)
        return outs[0], outs[1]
    K = dwt_coeff_len(N, L, mode)
    p = 2 * (K - 1) - N + L
    off = L - 1 - p // 2
    k = np.arange(K)
    for u in (u0, u1):
        y = 0
        for m in range(L):
            y = y + u[m] * _take_ext(x, 2 * k + off - m, mode, axis)
        outs.append(y)
    return outs[0], outs[1]


def sfb1d(lo, hi, g0, g1, mode='zero', axis=-1):
    """1-D synthesis bank along ``axis``.

    Restates dwt/lowlevel.py:226-271 (two conv_transpose2d, stride 2).  With K=len(lo):
      full[n] = sum_k lo[k] g0[n-2k] + hi[k] g1[n-2k],  n in [0, 2K+L-2)
      non-periodization (:262-269): y[n] = full[n+L-2],  n in [0, 2K-L+2)
      periodization (:252-261):     N=2K; full[:L-2] += full[N:N+L-2]; z = full[:N];
                                    y[i] = z[(i + S) mod N], S = L//2 - 1 (0 once L//2 - 1 >= 2N: roll() again)
    """
    if mode not in ('zero', 'symmetric', 'reflect', 'periodic', 'per', 'periodization'):
        raise ValueError("Unkown pad type: {}".format(mode))
    lo = np.asarray(lo)
    hi = np.asarray(hi)
    axis = axis % lo.ndim
    g0 = np.asarray(g0, dtype=lo.dtype).ravel()
    g1 = np.asarray(g1, dtype=lo.dtype).ravel()
    L = g0.size
    K = lo.shape[axis]
    nfull = 2 * K + L - 2
    shp = list(lo.shape)
    shp[axis] = nfull
    full = np.zeros(shp, dtype=lo.dtype)
    for t in range(L):
        sl = [slice(None)] * lo.ndim
        sl[axis] = slice(t, t + 2 * K, 2)
        full[tuple(sl)] += g0[t] * lo + g1[t] * hi
    if mode in ('per', 'periodization'):
        N = 2 * K
        nf = L - 2
        if nf > 0:
            sl_h = [slice(None)] * lo.ndim
            sl_h[axis] = slice(0, nf)
            sl_t = [slice(None)] * lo.ndim
            sl_t[axis] = slice(N, N + nf)
            # reference: y[:L-2] = y[:L-2] + y[N:N+L-2] on the FULL-length result (one fold only,
            # also when L-2 > N), then truncate to N
            full[tuple(sl_h)] = full[tuple(sl_h)] + full[tuple(sl_t)]
        z = np.take(full, np.arange(N), axis=axis)
        return np.take(z, np.mod(np.arange(N) + (L // 2 - 1 if L // 2 - 1 < 2 * N else 0), N), axis=axis)
    return np.take(full, np.arange(L - 2, L - 2 + 2 * K - L + 2), axis=axis)


# ----------------------------------------------------------------------------------------------
# DWT 2-D level + multi-level transforms
# ----------------------------------------------------------------------------------------------
def afb2d_level(x, h0_w, h1_w, h0_h, h1_h, mode):
    """One analysis level = AFB2D.forward (dwt/lowlevel.py:336-347): first pair filters along W
    (dim 3), second pair along H (dim 2).  Returns ``low (N,C,H',W')`` and
    ``highs (N,C,3,H',W')`` with band order [W-lo/H-hi ('LH'), W-hi/H-lo ('HL'), HH]."""
    lo, hi = afb1d(x, h0_w, h1_w, mode, axis=3)
    ll, lh = afb1d(lo, h0_h, h1_h, mode, axis=2)
    hl, hh = afb1d(hi, h0_h, h1_h, mode, axis=2)
    return ll, np.stack([lh, hl, hh], axis=2)


def sfb2d_level(low, highs, g0_w, g1_w, g0_h, g1_h, mode):
    """One synthesis level = SFB2D.forward (dwt/lowlevel.py:671-680): column (H) synthesis of
    (ll,lh) and (hl,hh) with the second pair, then row (W) synthesis with the first pair."""
    lh, hl, hh = highs[:, :, 0], highs[:, :, 1], highs[:, :, 2]
    lo = sfb1d(low, lh, g0_h, g1_h, mode, axis=2)
    hi = sfb1d(hl, hh, g0_h, g1_h, mode, axis=2)
    return sfb1d(lo, hi, g0_w, g1_w, mode, axis=3)


def afb2d_level_backward(dlow, dhighs, h0_w, h1_w, h0_h, h1_h, mode, in_shape):
    """AFB2D.backward (dwt/lowlevel.py:350-365): synthesis with the SAME stored analysis taps,
    then crop to the forward input's (H,W) (quirk Q9: not the exact adjoint for
    symmetric/reflect/periodic)."""
    dx = sfb2d_level(dlow, dhighs, h0_w, h1_w, h0_h, h1_h, mode)
    return dx[:, :, :in_shape[0], :in_shape[1]]


def sfb2d_level_backward(dy, g0_w, g1_w, g0_h, g1_h, mode):
    """SFB2D.backward (dwt/lowlevel.py:683-694): analysis with the stored synthesis taps."""
    return afb2d_level(dy, g0_w, g1_w, g0_h, g1_h, mode)


def dwt_forward(x, J, h0_col, h1_col, h0_row, h1_row, mode):
    """DWTForward.forward (dwt/transform2d.py:63-74).  Arguments are the module buffers in
    the module's order; NB the module hands them to AFB2D as (h0_col,h1_col,h0_row,h1_row) so
    the *col* pair filters along W and the *row* pair along H (quirk Q1)."""
    mode_to_int(mode)
    yh = []
    ll = np.asarray(x)
    for _ in range(J):
        ll, high = afb2d_level(ll, h0_col, h1_col, h0_row, h1_row, mode)
        yh.append(high)
    return ll, yh


def dwt_inverse(yl, yh, g0_col, g1_col, g0_row, g1_row, mode):
    """DWTInverse.forward (dwt/transform2d.py:131-148): coarsest first, ``None`` highs are
    zeros, ll is cropped by one row/col when it is larger than the highs."""
    mode_to_int(mode)
    ll = np.asarray(yl)
    for h in yh[::-1]:
        if h is None:
            h = np.zeros((ll.shape[0], ll.shape[1], 3, ll.shape[-2], ll.shape[-1]), dtype=ll.dtype)
        if ll.shape[-2] > h.shape[-2]:
            ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]:
            ll = ll[..., :-1]
        ll = sfb2d_level(ll, h, g0_col, g1_col, g0_row, g1_row, mode)
    return ll


# ----------------------------------------------------------------------------------------------
# DTCWT primitives (taps = stored/reversed buffers)
# ----------------------------------------------------------------------------------------------
def _corr_ext(X, hp, start, count, step, mode, axis):
    """Y[i] = sum_j hp[j] * ext(X, start + step*i + j) for i in [0,count)."""
    hp = np.asarray(hp, dtype=X.dtype).ravel()
    i = np.arange(count)
    y = 0
    for j in range(hp.size):
        y = y + hp[j] * _take_ext(X, start + step * i + j, mode, axis)
    return y


def colfilter(X, hp, mode='symmetric', axis=2):
    """dtcwt/lowlevel.py:70-80 (axis=2) / rowfilter :83-94 (axis=3).
    m=L//2;  Y[i] = sum_j hp[j] * ext(X, i + j - m),  i in [0, n+2m-L+1); ext is the symmetric
    (half-sample) extension for mode 'symmetric', zero padding for every other mode."""
    hp = np.asarray(hp).ravel()
    L = hp.size
    m = L // 2
    n = X.shape[axis]
    ext = 'symmetric' if mode == 'symmetric' else 'zero'
    return _corr_ext(X, hp, -m, n + 2 * m - L + 1, 1, ext, axis)


def rowfilter(X, hp, mode='symmetric'):
    return colfilter(X, hp, mode, axis=3)


def coldfilt(X, ha, hb, highpass=False, axis=2):
    """dtcwt/lowlevel.py:97-122 (axis=2) / rowdfilt :125-151 (axis=3).  n%4==0 required.
    m=L;  Ya[k] = sum_t ha[t] * sym(X, 4k+2t+2-m),  Yb[k] = sum_t hb[t] * sym(X, 4k+2t+3-m),
    k in [0,n/4);  Y[2k],Y[2k+1] = (Ya,Yb)[k]  (swapped when highpass)."""
    n = X.shape[axis]
    if n % 4 != 0:
        raise ValueError('No. of rows in X must be a multiple of 4\nX was {}'.format(X.shape))
    ha = np.asarray(ha, dtype=X.dtype).ravel()
    hb = np.asarray(hb, dtype=X.dtype).ravel()
    m = ha.size
    k = np.arange(n // 4)
    ya = 0
    yb = 0
    for t in range(m):
        ya = ya + ha[t] * _take_ext(X, 4 * k + 2 * t + 2 - m, 'symmetric', axis)
        yb = yb + hb[t] * _take_ext(X, 4 * k + 2 * t + 3 - m, 'symmetric', axis)
    first, second = (yb, ya) if highpass else (ya, yb)
    out = np.stack([first, second], axis=axis + 1)
    shp = list(X.shape)
    shp[axis] = n // 2
    return out.reshape(shp)


def rowdfilt(X, ha, hb, highpass=False):
    return coldfilt(X, ha, hb, highpass, axis=3)


def colifilt(X, ha, hb, highpass=False, axis=2):
    """dtcwt/lowlevel.py:154-195 (axis=2) / rowifilt :198-239 (axis=3).  n%2==0 required.
    m2=L/2, hao=ha[1::2], hae=ha[0::2] (stored order).
      Y[4q+s] = sum_{t<m2} f_s[t] * sym(X, o_s - m2 + 2(q+t)),  q in [0,n/2), s in 0..3
      m2 even: f=(hae,hbe,hao,hbo), o=(0,1,2,3)  [highpass: o=(1,0,3,2)]
      m2 odd : f=(hao,hbo,hae,hbe), o=(1,2,1,2)  [highpass: o=(2,1,2,1)]"""
    n = X.shape[axis]
    if n % 2 != 0:
        raise ValueError('No. of rows in X must be a multiple of 2.\nX was {}'.format(X.shape))
    ha = np.asarray(ha, dtype=X.dtype).ravel()
    hb = np.asarray(hb, dtype=X.dtype).ravel()
    m2 = ha.size // 2
    hao, hae, hbo, hbe = ha[1::2], ha[0::2], hb[1::2], hb[0::2]
    if m2 % 2 == 0:
        f = (hae, hbe, hao, hbo)
        o = (1, 0, 3, 2) if highpass else (0, 1, 2, 3)
    else:
        f = (hao, hbo, hae, hbe)
        o = (2, 1, 2, 1) if highpass else (1, 2, 1, 2)
    q = np.arange(n // 2)
    streams = []
    for s in range(4):
        y = 0
        for t in range(m2):
            y = y + f[s][t] * _take_ext(X, o[s] - m2 + 2 * (q + t), 'symmetric', axis)
        streams.append(y)
    out = np.stack(streams, axis=axis + 1)
    shp = list(X.shape)
    shp[axis] = 2 * n
    return out.reshape(shp)


def rowifilt(X, ha, hb, highpass=False):
    return colifilt(X, ha, hb, highpass, axis=3)


def q2c(y):
    """dtcwt/lowlevel.py:243-260: quads -> two complex sub-images ((z1r,z1i),(z2r,z2i))."""
    y = y / np.sqrt(2)
    a, b = y[:, :, 0::2, 0::2], y[:, :, 0::2, 1::2]
    c, d = y[:, :, 1::2, 0::2], y[:, :, 1::2, 1::2]
    return (a - d, b + c), (a + d, b - c)


def c2q(w1, w2):
    """dtcwt/lowlevel.py:263-295."""
    w1r, w1i = w1
    w2r, w2i = w2
    b, ch, r, c = w1r.shape
    y = np.zeros((b, ch, 2 * r, 2 * c), dtype=w1r.dtype)
    y[:, :, ::2, ::2] = w1r + w2r
    y[:, :, ::2, 1::2] = w1i + w2i
    y[:, :, 1::2, ::2] = w1i - w2i
    y[:, :, 1::2, 1::2] = -w1r + w2r
    return y / np.sqrt(2)


def highs_to_orientations(lh, hl, hh):
    """dtcwt/transform_funcs.py:61-72 with o_dim=2: returns reals, imags (N,C,6,h,w),
    orientation order [15,45,75,105,135,165] = [lh.z1, hh.z1, hl.z1, hl.z2, hh.z2, lh.z2]."""
    (d15r, d15i), (d165r, d165i) = q2c(lh)
    (d45r, d45i), (d135r, d135i) = q2c(hh)
    (d75r, d75i), (d105r, d105i) = q2c(hl)
    reals = np.stack([d15r, d45r, d75r, d105r, d135r, d165r], axis=2)
    imags = np.stack([d15i, d45i, d75i, d105i, d135i, d165i], axis=2)
    return reals, imags


def orientations_to_highs(reals, imags):
    """dtcwt/transform_funcs.py:75-95 with o_dim=2."""
    lh = c2q((reals[:, :, 0], imags[:, :, 0]), (reals[:, :, 5], imags[:, :, 5]))
    hl = c2q((reals[:, :, 2], imags[:, :, 2]), (reals[:, :, 3], imags[:, :, 3]))
    hh = c2q((reals[:, :, 1], imags[:, :, 1]), (reals[:, :, 4], imags[:, :, 4]))
    return lh, hl, hh


def fwd_j1(x, h0, h1, skip_hps=False, mode='symmetric'):
    """dtcwt/transform_funcs.py:98-121.  Returns ll and highs (N,C,6,h/2,w/2,2) (or None)."""
    lo = rowfilter(x, h0, mode)
    ll = colfilter(lo, h0, mode)
    if skip_hps:
        return ll, None
    hi = rowfilter(x, h1, mode)
    lh = colfilter(lo, h1, mode)
    hl = colfilter(hi, h0, mode)
    hh = colfilter(hi, h1, mode)
    r, i = highs_to_orientations(lh, hl, hh)
    return ll, np.stack([r, i], axis=-1)


def fwd_j2plus(x, h0a, h1a, h0b, h1b, skip_hps=False):
    """dtcwt/transform_funcs.py:226-249 (mode is always symmetric, :381)."""
    lo = rowdfilt(x, h0b, h0a, False)
    ll = coldfilt(lo, h0b, h0a, False)
    if skip_hps:
        return ll, None
    hi = rowdfilt(x, h1b, h1a, True)
    lh = coldfilt(lo, h1b, h1a, True)
    hl = coldfilt(hi, h0b, h0a, False)
    hh = coldfilt(hi, h1b, h1a, True)
    r, i = highs_to_orientations(lh, hl, hh)
    return ll, np.stack([r, i], axis=-1)


def inv_j1(ll, highs, g0, g1, mode='symmetric'):
    """dtcwt/transform_funcs.py:152-184.  ``highs`` (N,C,6,h,w,2) or None; ``ll`` or None."""
    if highs is None:
        # NB the reference ignores ``mode`` on this branch (:158-159)
        return rowfilter(colfilter(ll, g0), g0)
    lh, hl, hh = orientations_to_highs(highs[..., 0], highs[..., 1])
    hi = colfilter(hh, g1, mode) + colfilter(hl, g0, mode)
    lo = colfilter(lh, g1, mode)
    if ll is not None:
        r, c = ll.shape[2:]
        r1, c1 = highs.shape[3], highs.shape[4]
        if r != r1 * 2:
            ll = ll[:, :, 1:-1]
        if c != c1 * 2:
            ll = ll[:, :, :, 1:-1]
        lo = lo + colfilter(ll, g0, mode)
    return rowfilter(hi, g1, mode) + rowfilter(lo, g0, mode)


def inv_j2plus(ll, highs, g0a, g1a, g0b, g1b):
    """dtcwt/transform_funcs.py:279-307."""
    if highs is None:
        return rowifilt(colifilt(ll, g0b, g0a, False), g0b, g0a, False)
    lh, hl, hh = orientations_to_highs(highs[..., 0], highs[..., 1])
    hi = colifilt(hh, g1b, g1a, True) + colifilt(hl, g0b, g0a, False)
    lo = colifilt(lh, g1b, g1a, True)
    if ll is not None:
        lo = lo + colifilt(ll, g0b, g0a, False)
    return rowifilt(hi, g1b, g1a, True) + rowifilt(lo, g0b, g0a, False)


def dtcwt_forward(x, J, h0o, h1o, h0a, h0b, h1a, h1b, skip_hps=False, include_scale=False,
                  mode='symmetric'):
    """DTCWTForward.forward with o_dim=2, ri_dim=-1 (dtcwt/transform2d.py:87-147).
    Returns (yl or list of scales, [highs_j or None])."""
    x = np.asarray(x)
    skip = list(skip_hps) if isinstance(skip_hps, (list, tuple)) else [skip_hps] * J
    inc = list(include_scale) if isinstance(include_scale, (list, tuple)) else [include_scale] * J
    if J == 0:
        return x, None
    r, c = x.shape[2:]
    if r % 2 != 0:
        x = np.concatenate([x, x[:, :, -1:]], axis=2)
    if c % 2 != 0:
        x = np.concatenate([x, x[:, :, :, -1:]], axis=3)
    low, h = fwd_j1(x, h0o, h1o, skip[0], mode)
    highs = [h]
    scales = [low if inc[0] else None]
    for j in range(1, J):
        r, c = low.shape[2:]
        if r % 4 != 0:
            low = np.concatenate([low[:, :, 0:1], low, low[:, :, -1:]], axis=2)
        if c % 4 != 0:
            low = np.concatenate([low[:, :, :, 0:1], low, low[:, :, :, -1:]], axis=3)
        low, h = fwd_j2plus(low, h0a, h1a, h0b, h1b, skip[j])
        highs.append(h)
        scales.append(low if inc[j] else None)
    if True in inc:
        return scales, highs
    return low, highs


def dtcwt_inverse(yl, yh, g0o, g1o, g0a, g0b, g1a, g1b, mode='symmetric'):
    """DTCWTInverse.forward with o_dim=2, ri_dim=-1 (dtcwt/transform2d.py:193-254)."""
    low = yl
    J = len(yh)
    for j in range(J - 1, 0, -1):
        s = yh[j]
        if s is not None and low is not None:
            r, c = low.shape[2:]
            r1, c1 = s.shape[3], s.shape[4]
            if r != r1 * 2:
                low = low[:, :, 1:-1]
            if c != c1 * 2:
                low = low[:, :, :, 1:-1]
        low = inv_j2plus(low, s, g0a, g1a, g0b, g1b)
    s = yh[0]
    if s is not None and low is not None:
        r, c = low.shape[2:]
        r1, c1 = s.shape[3], s.shape[4]
        if r != r1 * 2:
            low = low[:, :, 1:-1]
        if c != c1 * 2:
            low = low[:, :, :, 1:-1]
    return inv_j1(low, s, g0o, g1o, mode)


# ----------------------------------------------------------------------------------------------
# ScatLayer (first order, non-rotational filters)
# ----------------------------------------------------------------------------------------------
def scat_layer_forward(x, h0o, h1o, mode='symmetric', magbias=1e-2, combine_colour=False,
                       return_saved=False):
    """ScatLayer.forward -> ScatLayerj1_f.forward (scatternet/layers.py:51-75,
    scatternet/lowlevel.py:76-111).  Returns (N,7C,H/2,W/2) (or (N,C+6,H/2,W/2) when
    combine_colour)."""
    x = np.asarray(x)
    _, ch, r, c = x.shape
    if r % 2 != 0:
        x = np.concatenate([x, x[:, :, -1:]], axis=2)
    if c % 2 != 0:
        x = np.concatenate([x, x[:, :, :, -1:]], axis=3)
    ll, highs = fwd_j1(x, h0o, h1o, False, mode)
    # fwd_j1 called with o_dim=1 upstream: (N,6,C,h,w)
    reals = np.moveaxis(highs[..., 0], 2, 1)
    imags = np.moveaxis(highs[..., 1], 2, 1)
    n, cc, H, W = ll.shape
    ll = ll.reshape(n, cc, H // 2, 2, W // 2, 2).mean(axis=(3, 5))
    if combine_colour:
        assert ch == 3
        rr = np.sqrt(reals[:, :, 0] ** 2 + imags[:, :, 0] ** 2 + reals[:, :, 1] ** 2 +
                     imags[:, :, 1] ** 2 + reals[:, :, 2] ** 2 + imags[:, :, 2] ** 2 + magbias ** 2)
        rr = rr[:, :, None]
    else:
        rr = np.sqrt(reals ** 2 + imags ** 2 + magbias ** 2)
    drdx, drdy = reals / rr, imags / rr
    rr = rr - magbias
    if combine_colour:
        Z = np.concatenate([ll, rr[:, :, 0]], axis=1)
    else:
        Z = np.concatenate([ll[:, None], rr], axis=1)
        b, _, c_, h_, w_ = Z.shape
        Z = Z.reshape(b, 7 * c_, h_, w_)
    if return_saved:
        return Z, (drdx, drdy)
    return Z


def scat_layer_backward(dZ, saved, h0o, h1o, mode='symmetric', combine_colour=False, C=None):
    """ScatLayerj1_f.backward (scatternet/lowlevel.py:114-137): dZ is (N,7C,h,w) (module
    view undone here) or (N,C+6,h,w) when combine_colour."""
    drdx, drdy = saved
    dZ = np.asarray(dZ)
    if combine_colour:
        dYl, dr = dZ[:, :3], dZ[:, 3:]
        dr = dr[:, :, None]
    else:
        b, c7, h, w = dZ.shape
        dZ = dZ.reshape(b, 7, c7 // 7, h, w)
        dYl, dr = dZ[:, 0], dZ[:, 1:]
    ll = 0.25 * np.repeat(np.repeat(dYl, 2, axis=2), 2, axis=3)
    reals = dr * drdx      # (N,6,C,h,w)
    imags = dr * drdy
    highs = np.stack([np.moveaxis(reals, 1, 2), np.moveaxis(imags, 1, 2)], axis=-1)
    return inv_j1(ll, highs, h0o, h1o, mode)


# ----------------------------------------------------------------------------------------------
# 1-D DWT and the undecimated (a-trous) bank
# ----------------------------------------------------------------------------------------------
def dwt1d_forward(x, J, h0, h1, mode):
    """DWT1DForward.forward (dwt/transform1d.py:38-59): J x AFB1D (dwt/lowlevel.py:368-407 = afb1d along the last
    axis).  x (N,C,L) -> (yl, [yh_0..])."""
    mode_to_int(mode)
    lo = np.asarray(x)[:, :, None, :]
    highs = []
    for _ in range(J):
        lo, hi = afb1d(lo, h0, h1, mode, axis=3)
        highs.append(hi[:, :, 0])
    return lo[:, :, 0], highs


def dwt1d_inverse(yl, yh, g0, g1, mode):
    """DWT1DInverse.forward (dwt/transform1d.py:93-115): coarsest first, None = zeros, 'unpad' by one sample."""
    mode_to_int(mode)
    x0 = np.asarray(yl)
    for x1 in yh[::-1]:
        if x1 is None:
            x1 = np.zeros_like(x0)
        if x0.shape[-1] > x1.shape[-1]:
            x0 = x0[..., :-1]
        x0 = sfb1d(x0[:, :, None, :], np.asarray(x1)[:, :, None, :], g0, g1, mode, axis=3)[:, :, 0]
    return x0


def _ext_any(idx, n, mode):
    """ext_index plus the two paddings only mypad knows (dwt/lowlevel.py:28-88): 'constant' (zeros) and 'replicate'."""
    if mode == 'constant':
        mode = 'zero'
    if mode == 'replicate':
        idx = np.asarray(idx, dtype=np.int64)
        return np.clip(idx, 0, n - 1), np.ones(idx.shape, dtype=bool)
    return ext_index(idx, n, mode)


def afb1d_atrous(x, h0, h1, mode='periodic', axis=-1, dilation=1):
    """dwt/lowlevel.py:175-223: pad (L*dil)//2 - dil before and (L*dil)//2 after with mypad(mode), then a dilated
    cross-correlation with the stored (reversed) taps:  y_b[i] = sum_t h_b[t] * ext(x, i - (L2 - dil) + dil*t).
    Returns (lo, hi), same length as x along `axis`."""
    if mode not in ('zero', 'constant', 'symmetric', 'reflect', 'periodic', 'replicate'):
        raise ValueError("Unkown pad type: {}".format(mode))
    x = np.asarray(x)
    axis = axis % x.ndim
    n = x.shape[axis]
    outs = []
    for h in (h0, h1):
        h = np.asarray(h, dtype=x.dtype).ravel()
        L = h.size
        L2 = (L * dilation) // 2
        K = n + 2 * L2 - dilation - dilation * (L - 1)
        i = np.arange(K)
        y = 0
        for t in range(L):
            src, valid = _ext_any(i - (L2 - dilation) + dilation * t, n, mode)
            v = np.take(x, src, axis=axis)
            if not valid.all():
                shp = [1] * x.ndim
                shp[axis] = -1
                v = v * valid.reshape(shp)
            y = y + h[t] * v
        outs.append(y)
    return outs[0], outs[1]


def afb2d_atrous(x, h0_col, h1_col, h0_row, h1_row, mode, dilation=1):
    """dwt/lowlevel.py:475-521: rows (the *row* pair along W) then columns; (N, 4C, H, W) with channel 4c + 2r + b."""
    lo, hi = afb1d_atrous(x, h0_row, h1_row, mode, 3, dilation)
    outs = []
    for r in (lo, hi):
        outs.extend(afb1d_atrous(r, h0_col, h1_col, mode, 2, dilation))
    n, c = x.shape[:2]
    return np.stack(outs, axis=2).reshape(n, 4 * c, outs[0].shape[2], outs[0].shape[3])


def scat_layer_j2_forward(x, h0o, h1o, h0a, h0b, h1a, h1b, magbias=1e-2, combine_colour=False):
    """ScatLayerj2.forward -> ScatLayerj2_f.forward (scatternet/layers.py:136-168, scatternet/lowlevel.py:205-295),
    symmetric mode (the only one whose second scale runs upstream).  Returns (N,49C,H/4,W/4), or (N,51,H/4,W/4) when
    combining colour."""
    x = np.asarray(x)
    ch, r, c = x.shape[1:]
    rem = r % 8
    if rem != 0:
        x = np.concatenate([x[:, :, :(8 - rem) // 2], x, x[:, :, -((9 - rem) // 2):]], axis=2)
    rem = c % 8
    if rem != 0:
        x = np.concatenate([x[:, :, :, :(8 - rem) // 2], x, x[:, :, :, -((9 - rem) // 2):]], axis=3)

    def pool(a):
        s = a.shape
        return a.reshape(s[:-2] + (s[-2] // 2, 2, s[-1] // 2, 2)).mean(axis=(-3, -1))

    def mags(highs, colour):
        re, im = np.moveaxis(highs[..., 0], 2, 1), np.moveaxis(highs[..., 1], 2, 1)   # (N,6,C,h,w)
        e = re ** 2 + im ** 2
        if colour:
            e = e.sum(axis=2, keepdims=True)
        return np.sqrt(e + magbias ** 2) - magbias

    s0, highs = fwd_j1(x, h0o, h1o, False, 'symmetric')
    s1_j1 = mags(highs, combine_colour)                     # (N,6,C or 1,H/2,W/2)
    s0, highs = fwd_j2plus(s0, h0a, h1a, h0b, h1b, False)
    s1_j2 = mags(highs, combine_colour)                     # (N,6,C or 1,H/4,W/4)
    s0 = pool(s0)
    n = x.shape[0]
    p = s1_j1.shape
    s1 = s1_j1.reshape(n, 6 * p[2], p[3], p[4])
    s1_ll, highs = fwd_j1(s1, h0o, h1o, False, 'symmetric')
    s2_j1 = mags(highs, False)                              # (N,6,6C',h,w)
    s1p = pool(s1_ll)
    h, w = s1p.shape[-2:]
    if combine_colour:
        return np.concatenate([s0, s1p, s1_j2[:, :, 0], s2_j1.reshape(n, 36, h, w)], axis=1)
    Z = np.concatenate([s0[:, None], s1p.reshape(n, 6, p[2], h, w), s1_j2, s2_j1.reshape(n, 36, p[2], h, w)], axis=1)
    return Z.reshape(n, 49 * p[2], h, w)


# ----------------------------------------------------------------------------------------------
# non-separable one-level banks (dwt/lowlevel.py:524-597, :746-798)
# ----------------------------------------------------------------------------------------------
def afb2d_nonsep(x, filts, mode='zero'):
    """x (N,C,H,W), filts (4,Ly,Lx) = the mirrored point-spread functions of prep_filt_afb2d_nonsep (:801-833) ->
    (N, 4C, H', W'), channel 4c+b.  The strided conv2d of :558-591 as a direct sum:
    zero / symmetric / reflect: y[b][i][j] = sum f[b][u][v] ext(x)[2i-(Ly-2)+u][2j-(Lx-2)+v] (p//2 = L-2 for every N, L);
    periodization: xe = x with the last row / column repeated for odd sizes (:559-564), rolled by ceil(L/2) (:567),
    zero-padded conv with padding L-1 and ONE fold of the first L//2 outputs (:568-570)."""
    x = np.asarray(x, dtype=np.float64)
    f = np.asarray(filts, dtype=np.float64).reshape(4, filts.shape[-2], filts.shape[-1])
    N, C, H, W = x.shape
    Ly, Lx = f.shape[1], f.shape[2]
    if mode in ('per', 'periodization'):
        if H % 2:
            x = np.concatenate((x, x[:, :, -1:]), axis=2)
        if W % 2:
            x = np.concatenate((x, x[:, :, :, -1:]), axis=3)
        He, We = x.shape[2], x.shape[3]
        xr = np.roll(np.roll(x, -((Ly + 1) // 2), axis=2), -((Lx + 1) // 2), axis=3)
        Ky, Kx = (He + Ly - 2) // 2 + 1, (We + Lx - 2) // 2 + 1
        xp = np.zeros((N, C, He + 2 * (Ly - 1), We + 2 * (Lx - 1)))
        xp[:, :, Ly - 1:Ly - 1 + He, Lx - 1:Lx - 1 + We] = xr
        y = np.zeros((N, C, 4, Ky, Kx))
        for u in range(Ly):
            for v in range(Lx):
                y += f[None, None, :, u, v, None, None] * xp[:, :, None, u:u + 2 * Ky:2, v:v + 2 * Kx:2]
        y[:, :, :, :Ly // 2] += y[:, :, :, He // 2:He // 2 + Ly // 2]
        y[:, :, :, :, :Lx // 2] += y[:, :, :, :, We // 2:We // 2 + Lx // 2]
        y = y[:, :, :, :He // 2, :We // 2]
    elif mode in ('zero', 'symmetric', 'reflect'):
        Ky, Kx = dwt_coeff_len(H, Ly, mode), dwt_coeff_len(W, Lx, mode)
        y = np.zeros((N, C, 4, Ky, Kx))
        for u in range(Ly):
            rows = _take_ext(x, 2 * np.arange(Ky) - (Ly - 2) + u, mode, 2)
            for v in range(Lx):
                y += f[None, None, :, u, v, None, None] * _take_ext(rows, 2 * np.arange(Kx) - (Lx - 2) + v, mode, 3)[:, :, None]
    else:
        raise ValueError("Unkown pad type: {}".format(mode))
    return y.reshape(N, 4 * C, y.shape[-2], y.shape[-1])


def sfb2d_nonsep(coeffs, filts, mode='zero'):
    """coeffs (N,C,4,H,W), filts (4,Ly,Lx) = prep_filt_sfb2d_nonsep (:836-867) -> (N,C,2H-Ly+2,2W-Lx+2), periodization
    (N,C,2H,2W): the conv_transpose2d of :784-796 as a direct sum (full[P][Q] = sum c[i][j] g[P-2i][Q-2j])."""
    c = np.asarray(coeffs, dtype=np.float64)
    g = np.asarray(filts, dtype=np.float64).reshape(4, filts.shape[-2], filts.shape[-1])
    N, C, _, Ny, Nx = c.shape
    Ly, Lx = g.shape[1], g.shape[2]
    full = np.zeros((N, C, 2 * Ny + Ly - 2, 2 * Nx + Lx - 2))
    for u in range(Ly):
        for v in range(Lx):
            full[:, :, u:u + 2 * Ny:2, v:v + 2 * Nx:2] += np.einsum('b,ncbij->ncij', g[:, u, v], c)
    if mode in ('per', 'periodization'):
        full[:, :, :Ly - 2] += full[:, :, 2 * Ny:2 * Ny + Ly - 2]
        full[:, :, :, :Lx - 2] += full[:, :, :, 2 * Nx:2 * Nx + Lx - 2]
        full = full[:, :, :2 * Ny, :2 * Nx]
        return np.roll(np.roll(full, 1 - Ly // 2, axis=2), 1 - Lx // 2, axis=3)
    if mode in ('zero', 'symmetric', 'reflect', 'periodic'):
        return full[:, :, Ly - 2:2 * Ny, Lx - 2:2 * Nx]
    raise ValueError("Unkown pad type: {}".format(mode))
