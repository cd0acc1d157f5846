"""Per-level DTCWT autograd Functions with the reference's names and call signatures
(pytorch_wavelets/dtcwt/transform_funcs.py:343-488); every forward / backward is one fused kernel
launch (level-1 / level>=2, forward / inverse) through the C ABI.

The kernels read / write the reference's DEFAULT coefficient layout (N, C, 6, H, W, 2); other
(o_dim, ri_dim) choices are a permutation of it.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops
from ..dwt.lowlevel import int_to_mode


def _is_empty(t):
    return t is None or t.shape == torch.Size([])


def get_dimensions5(o_dim, ri_dim):
    """Reference transform_funcs.py:10-29."""
    o_dim = o_dim % 6
    ri_dim = ri_dim % 6
    if ri_dim < o_dim:
        o_dim -= 1
    if o_dim == 4:
        h_dim, w_dim = 2, 3
    elif o_dim == 3:
        h_dim, w_dim = 2, 4
    else:
        h_dim, w_dim = 3, 4
    return o_dim, ri_dim, h_dim, w_dim


def get_dimensions6(o_dim, ri_dim):
    """Reference transform_funcs.py:32-58."""
    o_dim = o_dim % 6
    ri_dim = ri_dim % 6
    if ri_dim < o_dim:
        o_dim -= 1
    if o_dim >= 3 and ri_dim >= 3:
        h_dim = 2
    elif o_dim >= 4 or ri_dim >= 4:
        h_dim = 3
    else:
        h_dim = 4
    if o_dim >= 4 and ri_dim >= 4:
        w_dim = 3
    elif o_dim >= 4 or ri_dim >= 4:
        w_dim = 4
    else:
        w_dim = 5
    return o_dim, ri_dim, h_dim, w_dim


def _perm_from_default(o_dim, ri_dim):
    """Permutation p such that default (N,C,O,H,W,R) .permute(p) has O at o_dim and R at ri_dim."""
    o_dim, ri_dim = o_dim % 6, ri_dim % 6
    rest = iter([0, 1, 3, 4])      # N, C, H, W keep their relative order
    return [2 if d == o_dim else 5 if d == ri_dim else next(rest) for d in range(6)]


def to_layout(highs, o_dim, ri_dim):
    p = _perm_from_default(o_dim, ri_dim)
    return highs if p == [0, 1, 2, 3, 4, 5] else highs.permute(p).contiguous()


def from_layout(highs, o_dim, ri_dim):
    p = _perm_from_default(o_dim, ri_dim)
    if p == [0, 1, 2, 3, 4, 5]:
        return highs
    inv = [p.index(d) for d in range(6)]
    return highs.permute(inv)


def _unpad_grad_odd(dx, shape):
    """Backward of the edge replication that makes odd sizes even (transform2d.py:116-120)."""
    H, W = shape
    if dx.shape[2] > H:
        dx = torch.cat((dx[:, :, :H - 1], dx[:, :, H - 1:H] + dx[:, :, H:H + 1]), dim=2)
    if dx.shape[3] > W:
        dx = torch.cat((dx[:, :, :, :W - 1], dx[:, :, :, W - 1:W] + dx[:, :, :, W:W + 1]), dim=3)
    return dx


def _unpad_grad_both(dx, shape):
    """Backward of the one-row/column replication on BOTH sides (transform2d.py:131-135)."""
    H, W = shape
    if dx.shape[2] > H:
        mid = dx[:, :, 1:-1].clone()
        mid[:, :, 0] += dx[:, :, 0]
        mid[:, :, -1] += dx[:, :, -1]
        dx = mid
    if dx.shape[3] > W:
        mid = dx[:, :, :, 1:-1].clone()
        mid[:, :, :, 0] += dx[:, :, :, 0]
        mid[:, :, :, -1] += dx[:, :, :, -1]
        dx = mid
    return dx


class FWD_J1(Function):
    """Level-1 forward.  ``FWD_J1.apply(x, h0, h1, skip_hps, o_dim, ri_dim, mode_int) -> (ll, highs)``.
    Odd H/W are accepted: the kernel replicates the last row/column like DTCWTForward does upstream."""

    @staticmethod
    def forward(ctx, x, h0, h1, skip_hps, o_dim, ri_dim, mode):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(h0, h1)
        ctx.dims = (o_dim, ri_dim)
        ctx.in_hw = tuple(x.shape[-2:])
        ll, highs = ops.dtcwt_fwd1(x, h0, h1, mode, skip_hps)
        highs = ll.new_zeros([]) if skip_hps else to_layout(highs, o_dim, ri_dim)
        return ll, highs

    @staticmethod
    @once_differentiable
    def backward(ctx, dl, dh):
        dx = None
        if ctx.needs_input_grad[0]:
            h0, h1 = ctx.saved_tensors
            dh = None if _is_empty(dh) else from_layout(dh, *ctx.dims)
            dx = ops.dtcwt_inv1(dl, dh, h0, h1, ctx.mode)
            dx = _unpad_grad_odd(dx, ctx.in_hw)
        return dx, None, None, None, None, None, None


class FWD_J2PLUS(Function):
    """Level>=2 forward (always symmetric).  Backward = inv_j2plus with the a/b trees swapped
    (reference transform_funcs.py:395-413)."""

    @staticmethod
    def forward(ctx, x, h0a, h1a, h0b, h1b, skip_hps, o_dim, ri_dim, mode):
        ctx.save_for_backward(h0a, h1a, h0b, h1b)
        ctx.dims = (o_dim, ri_dim)
        ctx.in_hw = tuple(x.shape[-2:])
        if x.shape[-2] % 2 or x.shape[-1] % 2:
            raise ValueError('No. of rows in X must be a multiple of 4\nX was {}'.format(x.shape))
        ll, highs = ops.dtcwt_fwd2(x, h0a, h0b, h1a, h1b, skip_hps)
        highs = ll.new_zeros([]) if skip_hps else to_layout(highs, o_dim, ri_dim)
        return ll, highs

    @staticmethod
    @once_differentiable
    def backward(ctx, dl, dh):
        dx = None
        if ctx.needs_input_grad[0]:
            h0a, h1a, h0b, h1b = ctx.saved_tensors
            dh = None if _is_empty(dh) else from_layout(dh, *ctx.dims)
            # swapped trees: g0a := h0b, g0b := h0a, g1a := h1b, g1b := h1a
            dx = ops.dtcwt_inv2(dl, dh, h0b, h0a, h1b, h1a)
            dx = _unpad_grad_both(dx, ctx.in_hw)
        return dx, None, None, None, None, None, None, None, None


class INV_J1(Function):
    """Level-1 inverse.  ``INV_J1.apply(lows, highs, g0, g1, o_dim, ri_dim, mode_int) -> y``; ``lows`` or
    ``highs`` may be None / 0-dim (zeros)."""

    @staticmethod
    def forward(ctx, lows, highs, g0, g1, o_dim, ri_dim, mode):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(g0, g1)
        ctx.dims = (o_dim, ri_dim)
        ctx.has = (not _is_empty(lows), not _is_empty(highs))
        lows = None if _is_empty(lows) else lows
        highs = None if _is_empty(highs) else from_layout(highs, o_dim, ri_dim)
        return ops.dtcwt_inv1(lows, highs, g0, g1, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        g0, g1 = ctx.saved_tensors
        dl = dh = None
        need_l = ctx.has[0] and ctx.needs_input_grad[0]
        need_h = ctx.has[1] and ctx.needs_input_grad[1]
        if need_l or need_h:
            dl, dh = ops.dtcwt_fwd1(dy, g0, g1, ctx.mode, skip_hps=not need_h)
            dh = to_layout(dh, *ctx.dims) if need_h else None
            dl = dl if need_l else None
        return dl, dh, None, None, None, None, None


class INV_J2PLUS(Function):
    """Level>=2 inverse.  Backward = fwd_j2plus with the a/b trees swapped (reference :471-488)."""

    @staticmethod
    def forward(ctx, lows, highs, g0a, g1a, g0b, g1b, o_dim, ri_dim, mode):
        ctx.save_for_backward(g0a, g1a, g0b, g1b)
        ctx.dims = (o_dim, ri_dim)
        ctx.has = (not _is_empty(lows), not _is_empty(highs))
        lows = None if _is_empty(lows) else lows
        highs = None if _is_empty(highs) else from_layout(highs, o_dim, ri_dim)
        return ops.dtcwt_inv2(lows, highs, g0a, g0b, g1a, g1b)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        g0a, g1a, g0b, g1b = ctx.saved_tensors
        dl = dh = None
        need_l = ctx.has[0] and ctx.needs_input_grad[0]
        need_h = ctx.has[1] and ctx.needs_input_grad[1]
        if need_l or need_h:
            # swapped trees: h0a := g0b, h0b := g0a, h1a := g1b, h1b := g1a
            dl, dh = ops.dtcwt_fwd2(dy, g0b, g0a, g1b, g1a, skip_hps=not need_h)
            dh = to_layout(dh, *ctx.dims) if need_h else None
            dl = dl if need_l else None
        return dl, dh, None, None, None, None, None, None, None
