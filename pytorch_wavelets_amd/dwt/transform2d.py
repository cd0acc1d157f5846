"""DWTForward / DWTInverse with the reference's constructor signature, buffer names, (yl, yh)
layout and quirks (pytorch_wavelets/dwt/transform2d.py:7-148), running on the gfx950 engine."""
import torch
import torch.nn as nn

from .. import filters
from .. import ops
from . import lowlevel


def _resolve_bank(wave, lo_attr, hi_attr):
    """str | Wavelet-like | (f0, f1) | (f0_col, f1_col, f0_row, f1_row) -> four tap vectors."""
    if isinstance(wave, str):
        wave = filters.Wavelet(wave)
    if filters.is_wavelet_like(wave):
        c0, c1 = getattr(wave, lo_attr), getattr(wave, hi_attr)
        return c0, c1, c0, c1
    if len(wave) == 2:
        return wave[0], wave[1], wave[0], wave[1]
    if len(wave) == 4:
        return wave[0], wave[1], wave[2], wave[3]
    raise ValueError("wave must be a name, a Wavelet, or a tuple of 2 or 4 filters")


def _qmf_banks(g0_col, g1_col, g0_row, g1_row):
    """Are both highpass banks the quadrature mirrors of their lowpass banks?  (module-level: DWTInverse stays picklable)"""
    return ops.is_qmf_pair(g0_col, g1_col) and ops.is_qmf_pair(g0_row, g1_row)


def _same_banks(h0_col, h1_col, h0_row, h1_row):
    """Are the column bank and the row bank the same taps?  (module-level: DWTForward stays picklable)"""
    return ops.banks_equal(h0_col, h1_col, h0_row, h1_row)


class DWTForward(nn.Module):
    """2-D multi-level DWT.  ``DWTForward(J=1, wave='db1', mode='zero')(x) -> (yl, yh)`` with
    ``yh[j]`` of shape (N, C, 3, H_j, W_j), finest scale first (reference transform2d.py:7-74)."""

    def __init__(self, J=1, wave='db1', mode='zero'):
        super().__init__()
        h0_col, h1_col, h0_row, h1_row = _resolve_bank(wave, 'dec_lo', 'dec_hi')
        filts = lowlevel.prep_filt_afb2d(h0_col, h1_col, h0_row, h1_row)
        self.register_buffer('h0_col', filts[0])
        self.register_buffer('h1_col', filts[1])
        self.register_buffer('h0_row', filts[2])
        self.register_buffer('h1_row', filts[3])
        self.J = J
        self.mode = mode
        # kernel-variant hint, re-validated against the buffers on every call (see DWTInverse): the stored decomposition pair of
        # an orthogonal wavelet is a quadrature-mirror pair too, h1[t] = (-1)**t h0[L-1-t]
        self._qmf = ops.TapVerdict(_qmf_banks)
        # and: are the row and the column banks the same taps (one wavelet for both axes)?  The streaming kernel then keeps
        # one set of tap pairs in its scalar registers (ops.same_banks_hint)
        self._same = ops.TapVerdict(_same_banks)

    def forward(self, x):
        mode = lowlevel.mode_to_int(self.mode)
        if self.J < 1:
            return x, []
        # NB argument order: the module's *col* pair lands in the row slots (quirk Q1, reference
        # transform2d.py:70-71).  All J levels are one autograd node / (up to) one kernel launch.
        with ops.qmf_hint(self._qmf(self.h0_col, self.h1_col, self.h0_row, self.h1_row)), \
                ops.same_banks_hint(self._same(self.h0_col, self.h1_col, self.h0_row, self.h1_row)):
            outs = lowlevel.AFB2DMulti.apply(x, self.h0_col, self.h1_col, self.h0_row, self.h1_row, mode, self.J)
        return outs[0], list(outs[1:])


class DWTInverse(nn.Module):
    """2-D multi-level inverse DWT.  ``DWTInverse(wave='db1', mode='zero')((yl, yh)) -> x``;
    ``None`` entries of ``yh`` are zeros (reference transform2d.py:77-148)."""

    def __init__(self, wave='db1', mode='zero'):
        super().__init__()
        g0_col, g1_col, g0_row, g1_row = _resolve_bank(wave, 'rec_lo', 'rec_hi')
        filts = lowlevel.prep_filt_sfb2d(g0_col, g1_col, g0_row, g1_row)
        self.register_buffer('g0_col', filts[0])
        self.register_buffer('g1_col', filts[1])
        self.register_buffer('g0_row', filts[2])
        self.register_buffer('g1_row', filts[3])
        self.mode = mode
        # Kernel-variant hint, re-validated against the buffers on EVERY call (ops.TapVerdict): when each highpass bank is the
        # quadrature mirror of its lowpass bank, g1[t] = (-1)**t g0[L-1-t] (every orthogonal wavelet's reconstruction pair), the
        # streaming synthesis kernels derive the highpass tap pairs from the lowpass ones instead of holding both in scalar
        # registers (ops.qmf_hint).  The reference reads its buffers on every forward (transform2d.py:131-148): so does this.
        self._qmf = ops.TapVerdict(_qmf_banks)
        # and: one bank for both axes?  (with both hints the fused synthesis kernel runs its lattice variant, 10-20 taps)
        self._same = ops.TapVerdict(_same_banks)

    def forward(self, coeffs):
        yl, yh = coeffs
        mode = lowlevel.mode_to_int(self.mode)
        if len(yh) == 0:
            return yl
        with ops.qmf_hint(self._qmf(self.g0_col, self.g1_col, self.g0_row, self.g1_row)), \
                ops.same_banks_hint(self._same(self.g0_col, self.g1_col, self.g0_row, self.g1_row)):
            return lowlevel.SFB2DMulti.apply(yl, self.g0_col, self.g1_col, self.g0_row, self.g1_row, mode, *yh)


class SWTForward(nn.Module):
    """2-D stationary (undecimated) wavelet transform, ``pytorch_wavelets.dwt.transform2d.SWTForward`` (reference
    transform2d.py:151-212; not exported from the package upstream either).  One list entry per level, each
    (N, 4C, H, W) with the four sub-bands (ll, lh, hl, hh) of channel c at channels 4c..4c+3 - which is what the
    reference's level returns (its docstring promises (N, C, 4, H, W), its code does not reshape).

    Two upstream defects are NOT reproduced: the default ``mode='periodization'`` raises in upstream's ``mypad``
    (here too: use 'periodic', 'symmetric', 'reflect', 'zero', 'constant' or 'replicate'), and J > 1 crashes upstream
    (it indexes the 4-D level output as if it were 5-D); here level j+1 filters the ll channels of level j with the
    filters dilated by 2**j, which is the documented intent."""

    def __init__(self, J=1, wave='db1', mode='periodization'):
        super().__init__()
        h0_col, h1_col, h0_row, h1_row = _resolve_bank(wave, 'dec_lo', 'dec_hi')
        filts = lowlevel.prep_filt_afb2d(h0_col, h1_col, h0_row, h1_row)
        self.register_buffer('h0_col', filts[0])
        self.register_buffer('h1_col', filts[1])
        self.register_buffer('h0_row', filts[2])
        self.register_buffer('h1_row', filts[3])
        self.J = J
        self.mode = mode

    def forward(self, x):
        ll = x
        coeffs = []
        filts = (self.h0_col, self.h1_col, self.h0_row, self.h1_row)
        for j in range(self.J):
            y = lowlevel.afb2d_atrous(ll, filts, self.mode, 2 ** j)
            coeffs.append(y)
            ll = y[:, 0::4]
        return coeffs
