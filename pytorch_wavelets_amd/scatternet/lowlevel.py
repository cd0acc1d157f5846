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
