"""DTCWTForward / DTCWTInverse with the reference's constructor signatures, buffer names and (yl, yh)
layout (pytorch_wavelets/dtcwt/transform2d.py:20-254), running on the gfx950 engine."""
import numpy as np
import torch
import torch.nn as nn
from numpy import ndarray

from .. import ops
from ..dwt.lowlevel import mode_to_int
from ..filters import biort as _biort, qshift as _qshift
from .lowlevel import prep_filt
from .transform_funcs import FWD_J1, FWD_J12, FWD_J2PLUS, INV_J1, INV_J21, INV_J2PLUS, _perm_from_default


def _is_empty(t):
    return t is None or t.shape == torch.Size([])


class DTCWTForward(nn.Module):
    """2-D DTCWT.  ``DTCWTForward(biort='near_sym_a', qshift='qshift_a', J=3, skip_hps=False,
    include_scale=False, o_dim=2, ri_dim=-1, mode='symmetric')(x) -> (yl, yh)`` with ``yh[j]`` of
    shape (N, C, 6, H_j, W_j, 2) for the default o_dim / ri_dim (reference :20-147)."""

    def __init__(self, biort='near_sym_a', qshift='qshift_a', J=3, skip_hps=False, include_scale=False,
                 o_dim=2, ri_dim=-1, mode='symmetric'):
        super().__init__()
        if o_dim == ri_dim:
            raise ValueError("Orientations and real/imaginary parts must be in different dimensions.")
        self.biort, self.qshift, self.J = biort, qshift, J
        self.o_dim, self.ri_dim, self.mode = o_dim, ri_dim, mode
        if isinstance(biort, str):
            h0o, _, h1o, _ = _biort(biort)[:4]
        else:
            h0o, h1o = biort[0], biort[1]
        self.register_buffer('h0o', prep_filt(h0o, 1))
        self.register_buffer('h1o', prep_filt(h1o, 1))
        # (levels 1 and 2 run as one fused launch, FWD_J12, for ANY level-1 taps: the kernel meets the column lowpass taps in
        # reverse order on the rows it computes above / below the plane, which is exact whether or not h0o is symmetric - rounds
        # 3-4 required a symmetric h0o and checked the buffer on the host, a check that writes through `.data` escaped)
        if isinstance(qshift, str):
            h0a, h0b, _, _, h1a, h1b, _, _ = _qshift(qshift)[:8]
        else:
            h0a, h0b, h1a, h1b = qshift[0], qshift[1], qshift[2], qshift[3]
        self.register_buffer('h0a', prep_filt(h0a, 1))
        self.register_buffer('h0b', prep_filt(h0b, 1))
        self.register_buffer('h1a', prep_filt(h1a, 1))
        self.register_buffer('h1b', prep_filt(h1b, 1))
        self.skip_hps = skip_hps if isinstance(skip_hps, (list, tuple, ndarray)) else [skip_hps, ] * self.J
        self.include_scale = (include_scale if isinstance(include_scale, (list, tuple, ndarray))
                              else [include_scale, ] * self.J)

    def forward(self, x):
        mode = mode_to_int(self.mode)
        if self.J == 0:
            return x, None
        # (the reference fills both lists with 0-dim zeros up front, transform2d.py:107-108 - two fill launches per call; every
        # highs entry is assigned below, and the scales placeholders are only made when a scale is asked for)
        want_scales = True in self.include_scale
        scales = [x.new_zeros([]), ] * self.J if want_scales else None
        highs = [None, ] * self.J
        # odd sizes are extended by edge replication and sizes that are not multiples of 4 by one row /
        # column on both sides (reference :116-135): both happen inside the kernels
        first = 1
        if (self.J >= 2 and not self.skip_hps[0] and not self.skip_hps[1]
                and not self.include_scale[0]):
            low, highs[0], highs[1] = FWD_J12.apply(x, self.h0o, self.h1o, self.h0a, self.h1a, self.h0b, self.h1b,
                                                    self.o_dim, self.ri_dim, mode)
            if self.include_scale[1]:
                scales[1] = low
            first = 2
        else:
            low, h = FWD_J1.apply(x, self.h0o, self.h1o, self.skip_hps[0], self.o_dim, self.ri_dim, mode)
            highs[0] = h
            if self.include_scale[0]:
                scales[0] = low
        for j in range(first, self.J):
            low, h = FWD_J2PLUS.apply(low, self.h0a, self.h1a, self.h0b, self.h1b, self.skip_hps[j],
                                      self.o_dim, self.ri_dim, mode)
            highs[j] = h
            if self.include_scale[j]:
                scales[j] = low
        if want_scales:
            return scales, highs
        return low, highs


class DTCWTInverse(nn.Module):
    """2-D inverse DTCWT.  ``DTCWTInverse(biort, qshift, o_dim=2, ri_dim=-1, mode='symmetric')((yl, yh)) -> x``;
    entries of ``yh`` (and ``yl``) may be None / 0-dim tensors (reference :150-254)."""

    def __init__(self, biort='near_sym_a', qshift='qshift_a', o_dim=2, ri_dim=-1, mode='symmetric'):
        super().__init__()
        self.biort, self.qshift = biort, qshift
        self.o_dim, self.ri_dim, self.mode = o_dim, ri_dim, mode
        if isinstance(biort, str):
            _, g0o, _, g1o = _biort(biort)[:4]
        else:
            g0o, g1o = biort[0], biort[1]
        self.register_buffer('g0o', prep_filt(g0o, 1))
        self.register_buffer('g1o', prep_filt(g1o, 1))
        if isinstance(qshift, str):
            _, _, g0a, g0b, _, _, g1a, g1b = _qshift(qshift)[:8]
        else:
            g0a, g0b, g1a, g1b = qshift[0], qshift[1], qshift[2], qshift[3]
        self.register_buffer('g0a', prep_filt(g0a, 1))
        self.register_buffer('g0b', prep_filt(g0b, 1))
        self.register_buffer('g1a', prep_filt(g1a, 1))
        self.register_buffer('g1b', prep_filt(g1b, 1))

    def _crop_to(self, low, s, h_dim, w_dim):
        """Drop the 1-px border the forward transform added when a level was not a multiple of 4."""
        if _is_empty(low) or _is_empty(s):
            return low
        # (the reference crops once here and once more inside inv_j1 - transform_funcs.py:171-176 - which is
        # what makes pyramids with skipped levels line up: crop until the sizes agree)
        while low.shape[2] > s.shape[h_dim] * 2:
            low = low[:, :, 1:-1]
        while low.shape[3] > s.shape[w_dim] * 2:
            low = low[:, :, :, 1:-1]
        return low

    def forward(self, coeffs):
        low, highs = coeffs
        J = len(highs)
        mode = mode_to_int(self.mode)
        # positions of the H and W axes in the (o_dim, ri_dim) layout (the reference's get_dimensions6 is only
        # right for the common layouts; it matters solely for the crop-size comparison below)
        perm = _perm_from_default(self.o_dim, self.ri_dim)
        h_dim, w_dim = perm.index(3), perm.index(4)
        for j, s in zip(range(J - 1, 0, -1), highs[1:][::-1]):
            if j == 1 and not _is_empty(s) and not _is_empty(low) and not _is_empty(highs[0]):
                # the last two levels as one operator (one launch where the engine takes it) when no crop separates them
                low2 = self._crop_to(low, s, h_dim, w_dim)
                if (s.dim() == 6 and highs[0].dim() == 6 and s.shape[self.o_dim] == 6 and s.shape[self.ri_dim] == 2
                        and highs[0].shape[self.o_dim] == 6 and highs[0].shape[self.ri_dim] == 2   # (malformed level 1: the per-level path and its asserts)
                        and low2.shape[2] == 2 * s.shape[h_dim] and low2.shape[3] == 2 * s.shape[w_dim]
                        and low2.shape[2] == highs[0].shape[h_dim] and low2.shape[3] == highs[0].shape[w_dim]):
                    return INV_J21.apply(low2, s, highs[0], self.g0o, self.g1o, self.g0a, self.g1a, self.g0b, self.g1b,
                                         self.o_dim, self.ri_dim, mode)
            if not _is_empty(s):
                assert s.shape[self.o_dim] == 6, "Inverse transform must have input with 6 orientations"
                assert len(s.shape) == 6, "Bandpass inputs must have 6 dimensions"
                assert s.shape[self.ri_dim] == 2, \
                    "Inputs must be complex with real and imaginary parts in the ri dimension"
                low = self._crop_to(low, s, h_dim, w_dim)
            low = INV_J2PLUS.apply(low, s, self.g0a, self.g1a, self.g0b, self.g1b, self.o_dim, self.ri_dim, mode)
        low = self._crop_to(low, highs[0], h_dim, w_dim)
        return INV_J1.apply(low, highs[0], self.g0o, self.g1o, self.o_dim, self.ri_dim, mode)
