"""ScatLayer autograd Function (reference pytorch_wavelets/scatternet/lowlevel.py:71-137): forward is
ONE fused kernel (level-1 DTCWT + 2x2 LL average + smoothed magnitude, written straight into the
(N,7,C,h,w) output); backward is the level-1 inverse kernel fed with the re/r, im/r saved by the forward."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops
from ..dwt.lowlevel import int_to_mode, mode_to_int   # noqa: F401  (re-exported like upstream)


class ScatLayerj1_f(Function):
    @staticmethod
    def forward(ctx, x, h0o, h1o, mode, bias, combine_colour):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.combine_colour = combine_colour
        ctx.in_hw = tuple(x.shape[-2:])
        Z, drdx, drdy = ops.scat_fwd1(x, h0o, h1o, mode, bias, combine_colour, save=x.requires_grad)
        if x.requires_grad:
            ctx.save_for_backward(h0o, h1o, drdx, drdy)
        else:
            z = x.new_zeros(1)
            ctx.save_for_backward(h0o, h1o, z, z)
        return Z

    @staticmethod
    @once_differentiable
    def backward(ctx, dZ):
        dX = None
        if ctx.needs_input_grad[0]:
            h0o, h1o, drdx, drdy = ctx.saved_tensors
            dX = ops.scat_bwd1(dZ, drdx, drdy, h0o, h1o, ctx.mode, ctx.combine_colour)   # one fused launch
            if dX is None:   # no specialised kernel for these taps / dtype: prologue in torch + level-1 inverse
                if ctx.combine_colour:
                    dYl, dr = dZ[:, :3], dZ[:, 3:]
                    dr = dr[:, :, None]
                else:
                    dYl, dr = dZ[:, 0], dZ[:, 1:]
                ll = 0.25 * F.interpolate(dYl, scale_factor=2, mode="nearest")
                # (N,6,C,h,w) real / imag -> default coefficient layout (N,C,6,h,w,2)
                highs = torch.stack((dr * drdx, dr * drdy), dim=-1).permute(0, 2, 1, 3, 4, 5).contiguous()
                dX = ops.dtcwt_inv1(ll, highs, h0o, h1o, ctx.mode)
            H, W = ctx.in_hw
            if dX.shape[2] > H:   # gradient of the edge replication for odd sizes (layers.py:55-59 upstream)
                dX = torch.cat((dX[:, :, :H - 1], dX[:, :, H - 1:H] + dX[:, :, H:H + 1]), dim=2)
            if dX.shape[3] > W:
                dX = torch.cat((dX[:, :, :, :W - 1], dX[:, :, :, W - 1:W] + dX[:, :, :, W:W + 1]), dim=3)
        return (dX,) + (None,) * 5


class ScatLayerj1_ll_f(Function):
    """The first scale of ScatLayerj2_f (reference scatternet/lowlevel.py:214-262): ONE fused launch gives the
    full-resolution level-1 lowpass s0 AND the ScatLayer output Z (pooled lowpass + smoothed magnitudes).
    ``apply(x, h0o, h1o, mode_int, bias, combine_colour) -> (s0, Z)``.  Backward: the level-1 inverse is linear in
    (lowpass, highpasses), so d/dx = fused ScatLayer backward of dZ + level-1 inverse of ds0 alone - two launches."""

    @staticmethod
    def forward(ctx, x, h0o, h1o, mode, bias, combine_colour):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.combine_colour = combine_colour
        Z, drdx, drdy, ll = ops.scat_fwd1(x, h0o, h1o, mode, bias, combine_colour, save=x.requires_grad, want_ll=True)
        if x.requires_grad:
            ctx.save_for_backward(h0o, h1o, drdx, drdy)
        else:
            z = x.new_zeros(1)
            ctx.save_for_backward(h0o, h1o, z, z)
        return ll, Z

    @staticmethod
    @once_differentiable
    def backward(ctx, dll, dZ):
        dX = None
        if ctx.needs_input_grad[0]:
            h0o, h1o, drdx, drdy = ctx.saved_tensors
            dX = ops.scat_bwd1(dZ, drdx, drdy, h0o, h1o, ctx.mode, ctx.combine_colour)
            if dX is None:
                if ctx.combine_colour:
                    dYl, dr = dZ[:, :3], dZ[:, 3:]
                    dr = dr[:, :, None]
                else:
                    dYl, dr = dZ[:, 0], dZ[:, 1:]
                ll = 0.25 * F.interpolate(dYl, scale_factor=2, mode="nearest")
                highs = torch.stack((dr * drdx, dr * drdy), dim=-1).permute(0, 2, 1, 3, 4, 5).contiguous()
                dX = ops.dtcwt_inv1(ll + dll, highs, h0o, h1o, ctx.mode)
            else:
                dX = dX + ops.dtcwt_inv1(dll.contiguous(), None, h0o, h1o, ctx.mode)
        return (dX,) + (None,) * 5


def _smooth_mag(re, im, bias, sum_dim=None):
    """sqrt(re^2 + im^2 + b^2) - b (summed over colour first when combining), reference scatternet/lowlevel.py:225-262."""
    e = re * re + im * im
    if sum_dim is not None:
        e = e.sum(dim=sum_dim, keepdim=True)
    return torch.sqrt(e + bias * bias) - bias


FUSED_J2 = True    # inference: the three transforms of ScatLayerj2 write straight into its output (tests switch it off)


def _scat_layer_j2_in_place(x, h0o, h1o, h0a, h0b, h1a, h1b, mode, bias):
    """ScatLayerj2_f.forward without autograd and without colour combination: three launches, each writing its entries of the
    (N, 49, C, H/4, W/4) output in place (reference scatternet/lowlevel.py:205-295 builds them apart and concatenates):
      1. first scale: full-resolution lowpass s0 + first-order magnitudes (N, 6, C, H/2, W/2) (its pooled lowpass is not part
         of the output and is not written),
      2. second scale of s0: pooled lowpass -> entry 0, magnitudes -> entries 7..12,
      3. second-order layer on the 6C magnitude planes: pooled -> entries 1..6, its 36 magnitudes -> entries 13..48."""
    n, c, H, W = x.shape
    q2, q4 = (H // 2) * (W // 2), (H // 4) * (W // 4)
    Z = torch.empty((n, 49, c, H // 4, W // 4), dtype=x.dtype, device=x.device)
    m1 = torch.empty((n, 6 * c, H // 2, W // 2), dtype=x.dtype, device=x.device)   # first-order magnitudes, (N, 6, C, ..) order
    s0 = torch.empty((n, c, H, W), dtype=x.dtype, device=x.device)
    ops.scat_fwd1_into(x, m1, 6 * c * q2, -1, 0, h0o, h1o, mode, bias, ll=s0)
    if not ops.scat_fwd2_into(s0, Z, 49 * c * q4, 0, 7 * c * q4, h0a, h0b, h1a, h1b, bias):
        ll2, highs = ops.dtcwt_fwd2(s0, h0a, h0b, h1a, h1b)                        # highs (N,C,6,h,w,2)
        Z[:, 0] = F.avg_pool2d(ll2, 2)
        Z[:, 7:13] = _smooth_mag(highs[..., 0], highs[..., 1], bias).transpose(1, 2)
    ops.scat_fwd1_into(m1, Z, 49 * c * q4, c * q4, 13 * c * q4, h0o, h1o, mode, bias)
    return Z


def scat_layer_j2(x, h0o, h1o, h0a, h0b, h1a, h1b, mode, bias, combine_colour):
    """ScatLayerj2_f.forward (reference scatternet/lowlevel.py:205-295) as a chain of differentiable pieces, so that
    autograd reproduces its hand-written backward (:297-395; both are the exact adjoints - the level-1 filters are
    symmetric and the q-shift trees swap under time reversal):
      scale 1      : fused launch -> s0 (full size) + first-order magnitudes s1_j1          (ScatLayerj1_ll_f)
      scale 2      : level-2 DTCWT of s0 (FWD_J2PLUS) -> magnitudes s1_j2, 2x2 average of its lowpass
      second order : fused ScatLayer launch on s1_j1 -> its 2x2 average + 36 second-order magnitudes (ScatLayerj1_f)
    Returns (N, 49, C, H/4, W/4), or (N, 51, H/4, W/4) when combining colour."""
    from ..dtcwt.transform_funcs import FWD_J2PLUS
    if int_to_mode(mode) != 'symmetric':
        raise NotImplementedError()   # like upstream: the second scale's rowdfilt / coldfilt know only 'symmetric'
    if FUSED_J2 and not combine_colour and not (torch.is_grad_enabled() and x.requires_grad):
        return _scat_layer_j2_in_place(x, h0o, h1o, h0a, h0b, h1a, h1b, mode, bias)
    s0, Z1 = ScatLayerj1_ll_f.apply(x, h0o, h1o, mode, bias, combine_colour)
    ll2, highs = FWD_J2PLUS.apply(s0, h0a, h1a, h0b, h1b, False, 1, -1, mode)     # highs (N,6,C,h,w,2)
    s0 = F.avg_pool2d(ll2, 2)
    if combine_colour:
        s1_j1 = Z1[:, 3:]                                                         # (N,6,H/2,W/2)
        s1_j2 = _smooth_mag(highs[..., 0], highs[..., 1], bias, sum_dim=2)[:, :, 0]   # (N,6,h,w)
        Z2 = ScatLayerj1_f.apply(s1_j1, h0o, h1o, mode, bias, False)             # (N,7,6,h,w)
        n, _, _, h, w = Z2.shape
        return torch.cat((s0, Z2[:, 0], s1_j2, Z2[:, 1:].reshape(n, 36, h, w)), dim=1)
    n, _, c = Z1.shape[:3]
    s1_j1 = Z1[:, 1:].reshape(n, 6 * c, Z1.shape[3], Z1.shape[4])
    s1_j2 = _smooth_mag(highs[..., 0], highs[..., 1], bias)                       # (N,6,C,h,w)
    Z2 = ScatLayerj1_f.apply(s1_j1, h0o, h1o, mode, bias, False)                  # (N,7,6C,h,w)
    h, w = Z2.shape[-2:]
    return torch.cat((s0[:, None], Z2[:, 0].reshape(n, 6, c, h, w), s1_j2, Z2[:, 1:].reshape(n, 36, c, h, w)), dim=1)


def _scat_bwd1_any(dZ, drdx, drdy, h0o, h1o, mode, combine_colour=False):
    """ScatLayerj1_f.backward: the fused launch, or (taps / dtypes it has no kernel for) the prologue in the tensor library + the
    level-1 inverse."""
    dX = ops.scat_bwd1(dZ, drdx, drdy, h0o, h1o, mode, combine_colour)
    if dX is None:
        if combine_colour:
            dYl, dr = dZ[:, :3], dZ[:, 3:]
            dr = dr[:, :, None]
        else:
            dYl, dr = dZ[:, 0], dZ[:, 1:]
        ll = 0.25 * F.interpolate(dYl, scale_factor=2, mode="nearest")
        highs = torch.stack((dr * drdx, dr * drdy), dim=-1).permute(0, 2, 1, 3, 4, 5).contiguous()
        dX = ops.dtcwt_inv1(ll, highs, h0o, h1o, mode)
    return dX


class ScatLayerj1_rot_train_f(Function):
    """The TRAINING step of ScatLayerj1_rot_f (reference scatternet/lowlevel.py:140-203; no colour combination) on the fused
    ScatLayer kernels, two launches per direction (round 6).  fwd_j1_rot differs from fwd_j1 in ONE sub-band: hh = C_h2 R_h2 x
    instead of C_h1 R_h1 x (dtcwt/transform_funcs.py:124-149) - and the plain fused kernel run with the pair (h0o, h2o) computes
    exactly that as ITS hh.  So: launch A with (h0o, h1o) gives the pooled lowpass and the orientations of lh and hl (15, 75, 105,
    165 deg), launch B with (h0o, h2o) the orientations of hh (45, 135 deg: entries 2 and 5 of Z, 1 and 4 of the saved
    (re, im) / r).  The layer is a sum over sub-bands, so its backward is the fused backward A of dZ without those two entries
    plus the fused backward B of those two alone.  With the 13 / 19 / 19-tap tables all four launches are the streaming
    kernels of the 13 / 19 pair (WlDtFwd12Strip<T, 13, 19, 10, 3>, WlDtInv1Strip<T, 13, 19, 1>)."""
    # ``apply(x, h0o, h1o, h2o, mode_int, bias, want_ll) -> (ll, Z)``: ll = the full-resolution level-1 lowpass (ScatLayerj2's second
    # scale filters it; an empty tensor unless want_ll), Z (N, 7, C, H/2, W/2).  Also the inference form of that pair (no gradient
    # wanted: nothing is saved).
    @staticmethod
    def forward(ctx, x, h0o, h1o, h2o, mode, bias, want_ll=False, combine_colour=False):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.combine_colour = combine_colour
        ctx.e0 = e0 = 3 if combine_colour else 1              # entry of orientation 0: Z is (N, 3 + 6, h, w) when combining colour
        save = x.requires_grad
        res = ops.scat_fwd1(x, h0o, h1o, mode, bias, combine_colour, save=save, want_ll=want_ll)
        Z, drdx, drdy = res[:3]
        ll = res[3] if want_ll else x.new_zeros([])
        Zb, bx, by = ops.scat_fwd1(x, h0o, h2o, mode, bias, combine_colour, save=save)
        for o in (1, 4):
            Z[:, e0 + o] = Zb[:, e0 + o]
            if save:
                drdx[:, o] = bx[:, o]
                drdy[:, o] = by[:, o]
        if save:
            ctx.save_for_backward(h0o, h1o, h2o, drdx, drdy)
        else:
            z = x.new_zeros(1)
            ctx.save_for_backward(h0o, h1o, h2o, z, z)
        ctx.want_ll = want_ll
        return ll, Z

    @staticmethod
    @once_differentiable
    def backward(ctx, dll, dZ):
        dX = None
        if ctx.needs_input_grad[0]:
            h0o, h1o, h2o, drdx, drdy = ctx.saved_tensors
            dA = dZ.clone()
            dB = torch.zeros_like(dZ)
            for o in (1, 4):
                dB[:, ctx.e0 + o] = dZ[:, ctx.e0 + o]
                dA[:, ctx.e0 + o] = 0
            dX = _scat_bwd1_any(dA, drdx, drdy, h0o, h1o, ctx.mode, ctx.combine_colour) \
                + _scat_bwd1_any(dB, drdx, drdy, h0o, h2o, ctx.mode, ctx.combine_colour)
            if ctx.want_ll:    # the level-1 inverse is linear in (lowpass, highpasses): the lowpass's gradient alone
                dX = dX + ops.dtcwt_inv1(dll.contiguous(), None, h0o, h1o, ctx.mode)
        return (dX,) + (None,) * 7


ROT_TRAIN_FUSED = True   # tests switch it off to get the chain of differentiable pieces


def scat_layer_j1_rot(x, h0o, h1o, h2o, mode, bias, combine_colour):
    """ScatLayerj1_rot_f (reference scatternet/lowlevel.py:140-203): the ScatLayer with the rotationally symmetric
    13/19-tap filters (third band-pass pair for the diagonals), as a chain of differentiable pieces."""
    from ..dtcwt import transform_funcs as _tf
    from ..dtcwt.transform_funcs import FWD_J1_ROT
    if _tf.FUSED_ROT and not combine_colour and not (torch.is_grad_enabled() and x.requires_grad):
        # inference: the averaged lowpass and the magnitudes come out of the same launch (wl_dtcwt_fwd_level1_rot, scat = 1)
        z = ops.dtcwt_fwd1_rot(x, h0o, h1o, h2o, int_to_mode(mode) == 'symmetric', scat=True, magbias=bias)
        if z is not None:
            return z
    if ROT_TRAIN_FUSED and _tf.FUSED_ROT and x.shape[-2] % 2 == 0 and x.shape[-1] % 2 == 0 and h1o.numel() == h2o.numel():
        # training (and, when combining colour, inference): two launches of the fused ScatLayer kernels per direction (above)
        return ScatLayerj1_rot_train_f.apply(x, h0o, h1o, h2o, mode, bias, False, combine_colour)[1]
    ll, reals, imags = FWD_J1_ROT.apply(x, h0o, h1o, h2o, mode)
    ll = F.avg_pool2d(ll, 2)
    if combine_colour:
        r = _smooth_mag(reals, imags, bias, sum_dim=2)
        return torch.cat((ll, r[:, :, 0]), dim=1)
    return torch.cat((ll[:, None], _smooth_mag(reals, imags, bias)), dim=1)


def scat_layer_j2_rot(x, h0o, h1o, h2o, h0a, h0b, h1a, h1b, h2a, h2b, mode, bias, combine_colour):
    """ScatLayerj2_rot_f (reference scatternet/lowlevel.py:401-599), same chain as scat_layer_j2 on the band-pass
    level functions."""
    from ..dtcwt.transform_funcs import FWD_J1_ROT, FWD_J2PLUS_ROT, FWD_J2PLUS
    from ..dtcwt import transform_funcs as _tf
    if int_to_mode(mode) != 'symmetric':
        raise NotImplementedError()
    n = x.shape[0]
    if ROT_TRAIN_FUSED and _tf.FUSED_ROT and h1o.numel() == h2o.numel() and h1a.numel() == h2a.numel() and h1b.numel() == h2b.numel():
        # Round 6: every band-pass level function differs from the plain one in ONE sub-band - hh is filtered by the third pair on both
        # axes - and the plain fused kernels run with that pair in place of the highpass pair compute it as THEIR hh: each scale is two
        # launches of the plain kernels, the 45 / 135 degree orientations (entries 1 and 4) taken from the second (ScatLayerj1_rot_train_f).
        c = x.shape[1]
        s0, Z1 = ScatLayerj1_rot_train_f.apply(x, h0o, h1o, h2o, mode, bias, True, combine_colour)   # full-resolution lowpass; (N,7,C,H/2,W/2) | (N,9,H/2,W/2)
        s1_j1 = Z1[:, 3:] if combine_colour else Z1[:, 1:].reshape(n, 6 * c, Z1.shape[3], Z1.shape[4])
        ll2, ha = FWD_J2PLUS.apply(s0, h0a, h1a, h0b, h1b, False, 1, -1, mode)          # highs (N,6,C,h,w,2)
        _, hb = FWD_J2PLUS.apply(s0, h0a, h2a, h0b, h2b, False, 1, -1, mode)
        highs = torch.cat((ha[:, 0:1], hb[:, 1:2], ha[:, 2:4], hb[:, 4:5], ha[:, 5:6]), dim=1)
        s1_j2 = _smooth_mag(highs[..., 0], highs[..., 1], bias, sum_dim=2 if combine_colour else None)   # (N,6,C | 1,h,w)
        s0 = F.avg_pool2d(ll2, 2)
        Z2 = scat_layer_j1_rot(s1_j1, h0o, h1o, h2o, mode, bias, False)                  # (N,7,6C | 6,h,w)
        h, w = Z2.shape[-2:]
        if combine_colour:
            return torch.cat((s0, Z2[:, 0], s1_j2[:, :, 0], Z2[:, 1:].reshape(n, 36, h, w)), dim=1)
        return torch.cat((s0[:, None], Z2[:, 0].reshape(n, 6, c, h, w), s1_j2, Z2[:, 1:].reshape(n, 36, c, h, w)), dim=1)
    s0, reals, imags = FWD_J1_ROT.apply(x, h0o, h1o, h2o, mode)
    if combine_colour:
        s1_j1 = _smooth_mag(reals, imags, bias, sum_dim=2)[:, :, 0]                 # (N,6,H/2,W/2)
    else:
        s1_j1 = _smooth_mag(reals, imags, bias)                                     # (N,6,C,H/2,W/2)
        c = s1_j1.shape[2]
        s1_j1 = s1_j1.reshape(n, 6 * c, s1_j1.shape[3], s1_j1.shape[4])
    s0, reals, imags = FWD_J2PLUS_ROT.apply(s0, h0a, h1a, h0b, h1b, h2a, h2b, mode)
    s1_j2 = _smooth_mag(reals, imags, bias, sum_dim=2 if combine_colour else None)
    s0 = F.avg_pool2d(s0, 2)
    # second order: the first-order band-pass layer on the 6 C' magnitude planes - pooled lowpass + 36 magnitudes from ITS launches
    # (inference: one launch of the lean kernel or of WlDtFwd1Rot; training: ScatLayerj1_rot_train_f)
    if ROT_TRAIN_FUSED:
        Z2 = scat_layer_j1_rot(s1_j1, h0o, h1o, h2o, mode, bias, False)             # (N,7,6C',h,w)
        s1p, s2_j1 = Z2[:, 0], Z2[:, 1:]
    else:                                                                           # (the chain of round 4: A/B probes, tests)
        s1_ll, reals, imags = FWD_J1_ROT.apply(s1_j1, h0o, h1o, h2o, mode)
        s2_j1 = _smooth_mag(reals, imags, bias)                                     # (N,6,6C',h,w)
        s1p = F.avg_pool2d(s1_ll, 2)
    h, w = s1p.shape[-2:]
    if combine_colour:
        return torch.cat((s0, s1p, s1_j2[:, :, 0], s2_j1.reshape(n, 36, h, w)), dim=1)
    return torch.cat((s0[:, None], s1p.reshape(n, 6, c, h, w), s1_j2, s2_j1.reshape(n, 36, c, h, w)), dim=1)
