"""ScatLayer (reference pytorch_wavelets/scatternet/layers.py:11-79)."""
import torch
import torch.nn as nn

from ..dtcwt.lowlevel import prep_filt
from ..filters import biort as _biort, qshift as _qshift
from .lowlevel import ScatLayerj1_f, mode_to_int, scat_layer_j1_rot, scat_layer_j2, scat_layer_j2_rot


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
            # rotationally symmetric variant: a third (band-pass) pair filters the diagonal sub-band.  Inference: one launch (the lean
            # streaming kernel with a third row filter and window, or the tile kernel WlDtFwd1Rot); training: two launches of the plain
            # fused ScatLayer kernels per direction (scatternet/lowlevel.py: ScatLayerj1_rot_train_f), also with combine_colour
            self.bandpass_diag = True
            h0o, _, h1o, _, h2o, _ = _biort(biort)
            self.h2o = torch.nn.Parameter(prep_filt(h2o, 1), False)
        else:
            self.bandpass_diag = False
            h0o, _, h1o, _ = _biort(biort)[:4]
        self.h0o = torch.nn.Parameter(prep_filt(h0o, 1), False)
        self.h1o = torch.nn.Parameter(prep_filt(h1o, 1), False)

    def forward(self, x):
        _, ch, r, c = x.shape
        if self.combine_colour:
            assert ch == 3
        if self.bandpass_diag:
            if r % 2 != 0:   # replicate the last row / column (reference layers.py:55-59)
                x = torch.cat((x, x[:, :, -1:]), dim=2)
            if c % 2 != 0:
                x = torch.cat((x, x[:, :, :, -1:]), dim=3)
            Z = scat_layer_j1_rot(x, self.h0o, self.h1o, self.h2o, self.mode, self.magbias, self.combine_colour)
        else:
            # odd sizes are extended by edge replication inside the kernel (reference :55-59)
            Z = ScatLayerj1_f.apply(x, self.h0o, self.h1o, self.mode, self.magbias, self.combine_colour)
        if not self.combine_colour:
            b, _, c, h, w = Z.shape
            Z = Z.view(b, 7 * c, h, w)
        return Z

    def extra_repr(self):
        return "biort='{}', mode='{}', magbias={}".format(self.biort, self.mode_str, self.magbias)


class ScatLayerj2(nn.Module):
    """Second-order DTCWT scattering over two scales (reference scatternet/layers.py:82-172):
    ``ScatLayerj2(biort='near_sym_a', qshift='qshift_a', mode='symmetric', magbias=1e-2, combine_colour=False)(x)
    -> (N, 49C, H/4, W/4)`` (51 channels when combining colour): lowpass, 6 first-order terms of either scale and the
    36 second-order terms.  Inputs are extended to multiples of 8 like upstream."""

    def __init__(self, biort='near_sym_a', qshift='qshift_a', mode='symmetric', magbias=1e-2, combine_colour=False):
        super().__init__()
        self.biort = biort
        self.qshift = biort   # sic (upstream stores biort here)
        self.mode_str = mode
        self.mode = mode_to_int(mode)
        self.magbias = magbias
        self.combine_colour = combine_colour
        if biort == 'near_sym_b_bp':
            assert qshift == 'qshift_b_bp'
            self.bandpass_diag = True
            h0o, _, h1o, _, h2o, _ = _biort(biort)
            self.h2o = torch.nn.Parameter(prep_filt(h2o, 1), False)
            q = _qshift('qshift_b_bp')
            h0a, h0b, h1a, h1b, h2a, h2b = q[0], q[1], q[4], q[5], q[8], q[9]
            self.h2a = torch.nn.Parameter(prep_filt(h2a, 1), False)
            self.h2b = torch.nn.Parameter(prep_filt(h2b, 1), False)
        else:
            self.bandpass_diag = False
            h0o, _, h1o, _ = _biort(biort)[:4]
            h0a, h0b, _, _, h1a, h1b, _, _ = _qshift(qshift)[:8]
        self.h0o = torch.nn.Parameter(prep_filt(h0o, 1), False)
        self.h1o = torch.nn.Parameter(prep_filt(h1o, 1), False)
        self.h0a = torch.nn.Parameter(prep_filt(h0a, 1), False)
        self.h0b = torch.nn.Parameter(prep_filt(h0b, 1), False)
        self.h1a = torch.nn.Parameter(prep_filt(h1a, 1), False)
        self.h1b = torch.nn.Parameter(prep_filt(h1b, 1), False)

    def forward(self, x):
        ch, r, c = x.shape[1:]
        rem = r % 8
        if rem != 0:   # make the size a multiple of 8 by repeating border blocks (layers.py:138-150 upstream)
            rows_after = (9 - rem) // 2
            rows_before = (8 - rem) // 2
            x = torch.cat((x[:, :, :rows_before], x, x[:, :, -rows_after:]), dim=2)
        rem = c % 8
        if rem != 0:
            cols_after = (9 - rem) // 2
            cols_before = (8 - rem) // 2
            x = torch.cat((x[:, :, :, :cols_before], x, x[:, :, :, -cols_after:]), dim=3)
        if self.combine_colour:
            assert ch == 3
        if self.bandpass_diag:
            Z = scat_layer_j2_rot(x, self.h0o, self.h1o, self.h2o, self.h0a, self.h0b, self.h1a, self.h1b, self.h2a,
                                  self.h2b, self.mode, self.magbias, self.combine_colour)
        else:
            Z = scat_layer_j2(x, self.h0o, self.h1o, self.h0a, self.h0b, self.h1a, self.h1b, self.mode, self.magbias,
                              self.combine_colour)
        if not self.combine_colour:
            b, _, c, h, w = Z.shape
            Z = Z.reshape(b, 49 * c, h, w)
        return Z

    def extra_repr(self):
        return "biort='{}', mode='{}', magbias={}".format(self.biort, self.mode_str, self.magbias)
