"""DWT1DForward / DWT1DInverse with the reference's constructor signature, buffer names and (yl, yh) layout
(pytorch_wavelets/dwt/transform1d.py:7-115), on the engine's single-axis kernels (wl_corr1d / wl_synth1d)."""
import torch
import torch.nn as nn

from .. import filters
from . import lowlevel


def _resolve_pair(wave, lo_attr, hi_attr):
    if isinstance(wave, str):
        wave = filters.Wavelet(wave)
    if filters.is_wavelet_like(wave):
        return getattr(wave, lo_attr), getattr(wave, hi_attr)
    assert len(wave) == 2
    return wave[0], wave[1]


class DWT1DForward(nn.Module):
    """1-D multi-level DWT.  ``DWT1DForward(J=1, wave='db1', mode='zero')(x:(N,C,L)) -> (yl, [yh_0 ..])``, finest
    scale first (reference transform1d.py:7-59)."""

    def __init__(self, J=1, wave='db1', mode='zero'):
        super().__init__()
        h0, h1 = _resolve_pair(wave, 'dec_lo', 'dec_hi')
        filts = lowlevel.prep_filt_afb1d(h0, h1)
        self.register_buffer('h0', filts[0])
        self.register_buffer('h1', filts[1])
        self.J = J
        self.mode = mode

    def forward(self, x):
        assert x.ndim == 3, "Can only handle 3d inputs (N, C, L)"
        mode = lowlevel.mode_to_int(self.mode)
        if self.J < 1:
            return x, []
        # all J levels are one autograd node / (where the engine takes it) one kernel launch; more than four levels: in fours
        x0, highs = x, []
        left = self.J
        while left > 0:
            n = min(left, 4)
            outs = lowlevel.AFB1DMulti.apply(x0, self.h0, self.h1, mode, n)
            x0 = outs[0]
            highs.extend(outs[1:])
            left -= n
        return x0, highs


class DWT1DInverse(nn.Module):
    """1-D multi-level inverse DWT; ``None`` entries of ``yh`` are zeros (reference transform1d.py:62-115)."""

    def __init__(self, wave='db1', mode='zero'):
        super().__init__()
        g0, g1 = _resolve_pair(wave, 'rec_lo', 'rec_hi')
        filts = lowlevel.prep_filt_sfb1d(g0, g1)
        self.register_buffer('g0', filts[0])
        self.register_buffer('g1', filts[1])
        self.mode = mode

    def forward(self, coeffs):
        x0, highs = coeffs
        assert x0.ndim == 3, "Can only handle 3d inputs (N, C, L)"
        mode = lowlevel.mode_to_int(self.mode)
        highs = list(highs)
        # all levels are one autograd node / (where the engine takes it) one kernel launch; more than four levels: in fours,
        # coarsest first.  (A `None` level in a group: the per-level path inside the node, 'unpad' included.)
        while highs:
            grp = highs[-4:]
            highs = highs[:-4]
            x0 = lowlevel.SFB1DMulti.apply(x0, self.g0, self.g1, mode, *grp)
        return x0
