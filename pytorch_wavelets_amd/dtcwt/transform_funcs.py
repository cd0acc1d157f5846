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


class FWD_J12(Function):
    """Levels 1 and 2 of the forward as ONE operator: ``FWD_J12.apply(x, h0o, h1o, h0a, h1a, h0b, h1b, o_dim, ri_dim,
    mode_int) -> (ll2, highs1, highs2)`` = FWD_J1 followed by FWD_J2PLUS (reference dtcwt/transform2d.py:121-141) without
    the level-1 lowpass in between.  One launch of the fused streaming kernel where the engine takes it (any taps: nothing
    is assumed of ``h0o``), the two per-level launches otherwise; the backward is the chain of the two
    per-level backward passes (reference transform_funcs.py:361-374, :395-413)."""

    @staticmethod
    def forward(ctx, x, h0o, h1o, h0a, h1a, h0b, h1b, o_dim, ri_dim, mode):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(h0o, h1o, h0a, h1a, h0b, h1b)
        ctx.dims = (o_dim, ri_dim)
        ctx.in_hw = tuple(x.shape[-2:])
        res = ops.dtcwt_fwd12(x, h0o, h1o, h0a, h0b, h1a, h1b, mode)
        if res is None:
            ll1, highs1 = ops.dtcwt_fwd1(x, h0o, h1o, mode, False)
            ll2, highs2 = ops.dtcwt_fwd2(ll1, h0a, h0b, h1a, h1b, False)
            ctx.mid_hw = tuple(ll1.shape[-2:])
        else:
            highs1, ll2, highs2 = res
            ctx.mid_hw = ctx.in_hw
        return ll2, to_layout(highs1, o_dim, ri_dim), to_layout(highs2, o_dim, ri_dim)

    @staticmethod
    @once_differentiable
    def backward(ctx, dl2, dh1, dh2):
        dx = None
        if ctx.needs_input_grad[0]:
            h0o, h1o, h0a, h1a, h0b, h1b = ctx.saved_tensors
            dh1 = None if _is_empty(dh1) else from_layout(dh1, *ctx.dims)
            dh2 = None if _is_empty(dh2) else from_layout(dh2, *ctx.dims)
            dx = None
            if dh1 is not None and dh2 is not None and ctx.mid_hw == ctx.in_hw and tuple(dh1.shape[3:5]) == tuple(dl2.shape[2:]):
                # nothing was padded between the levels: one launch of the fused inverse (swapped trees, as below)
                dx = ops.dtcwt_inv21(dl2, dh2, dh1, h0o, h1o, h0b, h0a, h1b, h1a, ctx.mode)
            if dx is None:
                dl1 = ops.dtcwt_inv2(dl2, dh2, h0b, h0a, h1b, h1a)          # swapped trees, as in FWD_J2PLUS.backward
                dl1 = _unpad_grad_both(dl1, ctx.mid_hw)
                dx = ops.dtcwt_inv1(dl1, dh1, h0o, h1o, ctx.mode)
            dx = _unpad_grad_odd(dx, ctx.in_hw)
        return (dx,) + (None,) * 9


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


class INV_J21(Function):
    """Levels 2 and 1 of the inverse as ONE operator: ``INV_J21.apply(ll2, highs2, highs1, g0o, g1o, g0a, g1a, g0b, g1b, o_dim,
    ri_dim, mode_int) -> y`` = INV_J2PLUS followed by INV_J1 (reference dtcwt/transform2d.py:240-254) without the level-1 lowpass
    in between; all three inputs present and ``ll2`` exactly half the size of ``highs1``'s image (no crop between the levels).
    One launch of the fused streaming kernel where the engine takes it, the two per-level launches otherwise; the backward is
    the chain of the two per-level backward passes (reference transform_funcs.py:434-449, :471-488)."""

    @staticmethod
    def forward(ctx, ll2, highs2, highs1, g0o, g1o, g0a, g1a, g0b, g1b, o_dim, ri_dim, mode):
        int_to_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(g0o, g1o, g0a, g1a, g0b, g1b)
        ctx.dims = (o_dim, ri_dim)
        h2 = from_layout(highs2, o_dim, ri_dim)
        h1 = from_layout(highs1, o_dim, ri_dim)
        y = ops.dtcwt_inv21(ll2, h2, h1, g0o, g1o, g0a, g0b, g1a, g1b, mode)
        if y is None:
            y = ops.dtcwt_inv1(ops.dtcwt_inv2(ll2, h2, g0a, g0b, g1a, g1b), h1, g0o, g1o, mode)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        g0o, g1o, g0a, g1a, g0b, g1b = ctx.saved_tensors
        need = ctx.needs_input_grad
        dl2 = dh2 = dh1 = None
        if need[0] or need[1] or need[2]:
            need_2 = need[0] or need[1]
            dl1, dh1 = ops.dtcwt_fwd1(dy, g0o, g1o, ctx.mode, skip_hps=not need[2])
            dh1 = to_layout(dh1, *ctx.dims) if need[2] else None
            if need_2:
                # swapped trees, as in INV_J2PLUS.backward
                dl2, dh2 = ops.dtcwt_fwd2(dl1, g0b, g0a, g1b, g1a, skip_hps=not need[1])
                dh2 = to_layout(dh2, *ctx.dims) if need[1] else None
                dl2 = dl2 if need[0] else None
        return (dl2, dh2, dh1) + (None,) * 9


# ---------------------------------------------------------------------------------------------------------------------
# Rotationally symmetric variants ('near_sym_b_bp' / 'qshift_b_bp'): the diagonal sub-band is filtered by a third,
# band-pass pair h2 on both axes (reference transform_funcs.py:124-149, :187-223, :252-276, :310-340).  They are not on
# the fused hot path: each level is the reference's own decomposition into seven single-axis filters, every one a
# launch of the engine's correlation kernel (dtcwt/lowlevel.py here), plus q2c / c2q index shuffles.
# ---------------------------------------------------------------------------------------------------------------------
FUSED_ROT = True   # level 1 of the band-pass variants in one launch (False: the seven single-axis filters; A/B measurements)


def highs_to_orientations(lh, hl, hh, o_dim):
    """Three quad sub-bands -> six orientations 15..165 degrees, stacked along o_dim (reference :61-73)."""
    from .lowlevel import q2c
    (d15r, d15i), (d165r, d165i) = q2c(lh)
    (d45r, d45i), (d135r, d135i) = q2c(hh)
    (d75r, d75i), (d105r, d105i) = q2c(hl)
    return (torch.stack([d15r, d45r, d75r, d105r, d135r, d165r], dim=o_dim),
            torch.stack([d15i, d45i, d75i, d105i, d135i, d165i], dim=o_dim))


def orientations_to_highs(reals, imags, o_dim):
    """Inverse of highs_to_orientations (reference :76-96)."""
    from .lowlevel import c2q
    r = torch.unbind(reals, dim=o_dim)
    i = torch.unbind(imags, dim=o_dim)
    lh = c2q((r[0], i[0]), (r[5], i[5]))
    hl = c2q((r[2], i[2]), (r[3], i[3]))
    hh = c2q((r[1], i[1]), (r[4], i[4]))
    return lh, hl, hh


def fwd_j1_rot(x, h0, h1, h2, skip_hps, o_dim, mode):
    """Level-1 forward with the band-pass diagonal (reference :124-149).  mode: 'symmetric' or anything else = zero."""
    from .lowlevel import colfilter, rowfilter
    if skip_hps:
        return colfilter(rowfilter(x, h0, mode), h0, mode), x.new_zeros([]), x.new_zeros([])
    if o_dim == 1 and FUSED_ROT:
        res = ops.dtcwt_fwd1_rot(x, h0, h1, h2, mode == 'symmetric')      # one launch (csrc/wl_dtcwt_rot.h)
        if res is not None:
            return res
    lo = rowfilter(x, h0, mode)
    hi = rowfilter(x, h1, mode)
    ba = rowfilter(x, h2, mode)
    lh = colfilter(lo, h1, mode)
    hl = colfilter(hi, h0, mode)
    hh = colfilter(ba, h2, mode)
    ll = colfilter(lo, h0, mode)
    highr, highi = highs_to_orientations(lh, hl, hh, o_dim)
    return ll, highr, highi


def inv_j1_rot(ll, highr, highi, g0, g1, g2, o_dim, h_dim, w_dim, mode):
    """Level-1 inverse with the band-pass diagonal (reference :187-223)."""
    from .lowlevel import colfilter, rowfilter
    if _is_empty(highr):
        return rowfilter(colfilter(ll, g0), g0)
    lh, hl, hh = orientations_to_highs(highr, highi, o_dim)
    if _is_empty(ll):
        lo = colfilter(lh, g1, mode)
    else:
        r, c = ll.shape[2:]
        if r != highr.shape[h_dim] * 2:
            ll = ll[:, :, 1:-1]
        if c != highr.shape[w_dim] * 2:
            ll = ll[:, :, :, 1:-1]
        lo = colfilter(lh, g1, mode) + colfilter(ll, g0, mode)
    hi = colfilter(hl, g0, mode)
    ba = colfilter(hh, g2, mode)
    return rowfilter(hi, g1, mode) + rowfilter(lo, g0, mode) + rowfilter(ba, g2, mode)


def fwd_j2plus_rot(x, h0a, h1a, h0b, h1b, h2a, h2b, skip_hps, o_dim, mode):
    """Level >= 2 forward with the band-pass diagonal (reference :252-276)."""
    from .lowlevel import coldfilt, rowdfilt
    if skip_hps:
        return coldfilt(rowdfilt(x, h0b, h0a, False, mode), h0b, h0a, False, mode), None, None
    lo = rowdfilt(x, h0b, h0a, False, mode)
    hi = rowdfilt(x, h1b, h1a, True, mode)
    ba = rowdfilt(x, h2b, h2a, True, mode)
    lh = coldfilt(lo, h1b, h1a, True, mode)
    hl = coldfilt(hi, h0b, h0a, False, mode)
    hh = coldfilt(ba, h2b, h2a, True, mode)
    ll = coldfilt(lo, h0b, h0a, False, mode)
    highr, highi = highs_to_orientations(lh, hl, hh, o_dim)
    return ll, highr, highi


def inv_j2plus_rot(ll, highr, highi, g0a, g1a, g0b, g1b, g2a, g2b, o_dim, h_dim, w_dim, mode):
    """Level >= 2 inverse with the band-pass diagonal (reference :310-340)."""
    from .lowlevel import colifilt, rowifilt
    if _is_empty(highr):
        return rowifilt(colifilt(ll, g0b, g0a, False, mode), g0b, g0a, False, mode)
    lh, hl, hh = orientations_to_highs(highr, highi, o_dim)
    lo = colifilt(lh, g1b, g1a, True, mode)
    if not _is_empty(ll):
        lo = lo + colifilt(ll, g0b, g0a, False, mode)
    hi = colifilt(hl, g0b, g0a, False, mode)
    ba = colifilt(hh, g2b, g2a, True, mode)
    return rowifilt(hi, g1b, g1a, True, mode) + rowifilt(lo, g0b, g0a, False, mode) + rowifilt(ba, g2b, g2a, True, mode)


class FWD_J1_ROT(Function):
    """``FWD_J1_ROT.apply(x, h0, h1, h2, mode_int) -> (ll, reals, imags)``, orientations along dim 1 ((N,6,C,h,w)).
    Backward = inv_j1_rot with the SAME filters (they are symmetric), as ScatLayerj1_rot_f.backward does upstream
    (scatternet/lowlevel.py:184-203)."""

    @staticmethod
    def forward(ctx, x, h0, h1, h2, mode):
        ctx.mode = int_to_mode(mode)
        ctx.save_for_backward(h0, h1, h2)
        return fwd_j1_rot(x, h0, h1, h2, False, 1, ctx.mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dll, dre, dim):
        dx = None
        if ctx.needs_input_grad[0]:
            h0, h1, h2 = ctx.saved_tensors
            dx = inv_j1_rot(dll, dre, dim, h0, h1, h2, 1, 3, 4, ctx.mode)
        return dx, None, None, None, None


class FWD_J2PLUS_ROT(Function):
    """``FWD_J2PLUS_ROT.apply(x, h0a, h1a, h0b, h1b, h2a, h2b, mode_int) -> (ll, reals, imags)``.  Backward =
    inv_j2plus_rot with the a / b trees swapped (time reversal; scatternet/lowlevel.py:513-521 upstream)."""

    @staticmethod
    def forward(ctx, x, h0a, h1a, h0b, h1b, h2a, h2b, mode):
        ctx.mode = int_to_mode(mode)
        ctx.save_for_backward(h0a, h1a, h0b, h1b, h2a, h2b)
        return fwd_j2plus_rot(x, h0a, h1a, h0b, h1b, h2a, h2b, False, 1, ctx.mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dll, dre, dim):
        dx = None
        if ctx.needs_input_grad[0]:
            h0a, h1a, h0b, h1b, h2a, h2b = ctx.saved_tensors
            dx = inv_j2plus_rot(dll, dre, dim, h0b, h1b, h0a, h1a, h2b, h2a, 1, 3, 4, ctx.mode)
        return (dx,) + (None,) * 7
