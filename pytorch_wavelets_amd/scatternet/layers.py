"""ScatLayer (reference pytorch_wavelets/scatternet/layers.py:11-79)."""
import torch
import torch.nn as nn

from ..dtcwt.lowlevel import prep_filt
from ..filters import biort as _biort
from .lowlevel import ScatLayerj1_f, mode_to_int


class ScatLayer(nn.Module):
    """One order of DTCWT scattering at one scale: ``ScatLayer(biort='near_sym_a', mode='symmetric',
    magbias=1e-2, combine_colour=False)(x) -> (N, 7C, ceil(H/2), ceil(W/2))`` with channel order
    [lowpass x C, 15deg x C, 45deg x C, ..., 165deg x C]."""

    def __init__(self, biort='near_sym_a', mode='symmetric', magbias=1e-2, combine_colour=False):
        super().__init__()
        self.biort = biort
        self.mode_str = mode
        self.mode = mode_to_int(mode)
        self.magbias = magbias
        self.combine_colour = combine_colour
        if biort == 'near_sym_b_bp':
            raise NotImplementedError("the rotationally symmetric band-pass variant ('near_sym_b_bp') is not "
                                      "implemented by the gfx950 engine yet")
        self.bandpass_diag = False
        h0o, _, h1o, _ = _biort(biort)[:4]
        self.h0o = torch.nn.Parameter(prep_filt(h0o, 1), False)
        self.h1o = torch.nn.Parameter(prep_filt(h1o, 1), False)

    def forward(self, x):
        _, ch, r, c = x.shape
        if self.combine_colour:
            assert ch == 3
        # odd sizes are extended by edge replication inside the kernel (reference :55-59)
        Z = ScatLayerj1_f.apply(x, self.h0o, self.h1o, self.mode, self.magbias, self.combine_colour)
        if not self.combine_colour:
            b, _, c, h, w = Z.shape
            Z = Z.view(b, 7 * c, h, w)
        return Z

    def extra_repr(self):
        return "biort='{}', mode='{}', magbias={}".format(self.biort, self.mode_str, self.magbias)
