"""Per-scale DWT operators (layer L2 of SURVEY.md): the autograd Functions the reference exposes in
``pytorch_wavelets/dwt/lowlevel.py``, with the same names, argument order, mode codes and error
text - but each forward/backward is ONE fused gfx950 kernel launch through the C ABI instead of
a chain of ATen gathers / grouped convs / copies.
"""
import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops

_MODE_TO_INT = {'zero': 0, 'symmetric': 1, 'per': 2, 'periodization': 2, 'constant': 3, 'reflect': 4,
                'replicate': 5, 'periodic': 6}
_INT_TO_MODE = {0: 'zero', 1: 'symmetric', 2: 'periodization', 3: 'constant', 4: 'reflect',
                5: 'replicate', 6: 'periodic'}
FUSED_LEVELS = True   # set False to force one launch per level (A/B measurements)
WIDE_ONE_LEVEL = 640  # coefficient-row pairs: a synthesis level left on its own whose output is at least this wide goes to the one-level strip kernel (SFB2DMulti.forward)
_FILTERBANK_MODES = (0, 1, 2, 4, 6)   # the ones afb1d/sfb1d accept upstream (dwt/lowlevel.py:134-170)


def mode_to_int(mode):
    """Reference dwt/lowlevel.py:274-290."""
    try:
        return _MODE_TO_INT[mode]
    except (KeyError, TypeError):
        raise ValueError("Unkown pad type: {}".format(mode))


def int_to_mode(mode):
    """Reference dwt/lowlevel.py:293-309."""
    try:
        return _INT_TO_MODE[mode]
    except (KeyError, TypeError):
        raise ValueError("Unkown pad type: {}".format(mode))


def _check_bank_mode(mode):
    if mode not in _FILTERBANK_MODES:
        raise ValueError("Unkown pad type: {}".format(int_to_mode(mode)))


class AFB2D(Function):
    """One level of 2-D analysis.  ``AFB2D.apply(x, h0_row, h1_row, h0_col, h1_col, mode_int)
    -> (low (N,C,H',W'), highs (N,C,3,H',W'))``; the *row* pair filters along W, the *col* pair
    along H (reference dwt/lowlevel.py:336-347).  Backward = synthesis with the same stored taps,
    cropped to the input size (reference :350-365, quirk Q9 reproduced)."""

    @staticmethod
    def forward(ctx, x, h0_row, h1_row, h0_col, h1_col, mode):
        _check_bank_mode(mode)
        ctx.save_for_backward(h0_row, h1_row, h0_col, h1_col)
        ctx.shape = x.shape[-2:]
        ctx.mode = mode
        return ops.afb2d_best(x, h0_row, h1_row, h0_col, h1_col, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, low, highs):
        dx = None
        if ctx.needs_input_grad[0]:
            h0_row, h1_row, h0_col, h1_col = ctx.saved_tensors
            dx = ops.sfb2d_best(low, highs, h0_row, h1_row, h0_col, h1_col, ctx.mode,
                           out_hw=tuple(ctx.shape))
        return dx, None, None, None, None, None


class SFB2D(Function):
    """One level of 2-D synthesis.  ``SFB2D.apply(low, highs, g0_row, g1_row, g0_col, g1_col,
    mode_int) -> y`` (reference dwt/lowlevel.py:671-680).  Backward = analysis with the stored
    synthesis taps (reference :683-694)."""

    @staticmethod
    def forward(ctx, low, highs, g0_row, g1_row, g0_col, g1_col, mode):
        _check_bank_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(g0_row, g1_row, g0_col, g1_col)
        ctx.has_highs = highs is not None
        return ops.sfb2d_best(low, highs, g0_row, g1_row, g0_col, g1_col, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dlow, dhigh = None, None
        if ctx.needs_input_grad[0] or (ctx.has_highs and ctx.needs_input_grad[1]):
            g0_row, g1_row, g0_col, g1_col = ctx.saved_tensors
            dlow, dhigh = ops.afb2d_best(dy, g0_row, g1_row, g0_col, g1_col, ctx.mode)
            if not ctx.has_highs:
                dhigh = None
        return dlow, dhigh, None, None, None, None, None


class SFB2DMulti(Function):
    """All synthesis levels as ONE autograd node: ``SFB2DMulti.apply(yl, g0_row, g1_row, g0_col, g1_col, mode_int,
    *yh) -> x`` with yh finest first, ``None`` entries = zeros (the level loop of DWTInverse.forward, reference
    dwt/transform2d.py:131-148, incl. the 'unpad' of a low-pass one row / column larger than the next high-pass).

    Forward: up to three levels at a time in ONE launch of the streaming kernel (wl_dwt2d_synthesis_fused: the
    intermediate low-passes stay in LDS) whenever the engine takes the configuration, otherwise one tile-kernel launch
    per level.  Backward = the chain of SFB2D.backward steps of the reference (analysis with the stored synthesis
    taps, dwt/lowlevel.py:683-694), finest level first; a dropped row / column gets a zero gradient."""

    @staticmethod
    def forward(ctx, yl, g0_row, g1_row, g0_col, g1_col, mode, *yh):
        _check_bank_mode(mode)
        ctx.save_for_backward(g0_row, g1_row, g0_col, g1_col)
        ctx.mode = mode
        ctx.hints = ops.current_hints()     # (the module's kernel-variant hints, re-installed around the backward pass: autograd's thread)
        ctx.has_highs = [h is not None for h in yh]
        J = len(yh)
        ll_shapes = [None] * J          # the low-pass handed to level j, before the 'unpad'
        ll, j = yl, J - 1
        while j >= 0:
            n = 0                       # levels j, j-1, .. that one launch can take together
            while n < 4 and j - n >= 0 and yh[j - n] is not None:
                n += 1
            # small planes (CNN feature maps): several planes per workgroup, up to four levels in LDS
            res = ops.sfb2d_small(ll, list(yh[j - n + 1:j + 1]), g0_row, g1_row, g0_col, g1_col, mode) if FUSED_LEVELS and n else None
            if res is None and n:
                # More levels than one streaming launch takes (three): the COARSEST (j mod 3) + 1 go first, so that the finest -
                # nearly all of the bytes - go three to a launch (J = 4 as 3 + 1 from the coarse end ran its finest level
                # alone: 0.288 ms against 0.188 for J = 3 at 128 x 3 x 512 x 512); the coarse remainder are small planes.
                m = min(n, 3, (j % 3) + 1 if j + 1 > 3 else 3)
                if FUSED_LEVELS and m < n:
                    res = ops.sfb2d_small(ll, list(yh[j - m + 1:j + 1]), g0_row, g1_row, g0_col, g1_col, mode)
                n = m
            took_strip = False
            while FUSED_LEVELS and n >= 1 and res is None:
                if n == 1 and 2 * yh[j].shape[-1] >= WIDE_ONE_LEVEL:
                    # a single WIDE level: the one-level strip kernel is ahead of the fused kernel's one-level form (same-box, float32,
                    # tools/gpu_r5u.py: 64x3x1024^2 0.333 -> 0.309 ms, 128x3x768^2 0.362 -> 0.342, 128x3x640^2 0.244 -> 0.228);
                    # when it declines (few planes, a width that is no multiple of four) the fused kernel is asked as before
                    # (the reference's 'unpad' drops exactly ONE surplus row / column, dwt/transform2d.py:141-146: a low-pass that is
                    # larger than that is malformed and takes the per-level path below, which raises like every other path)
                    h = yh[j]
                    dh, dw = ll.shape[-2] - h.shape[-2], ll.shape[-1] - h.shape[-1]
                    lc = ll[..., :h.shape[-2], :h.shape[-1]]
                    one = (ops.sfb2d_stream(lc, h, g0_row, g1_row, g0_col, g1_col, mode, force=ops.STREAM_FORCE)
                           if 0 <= dh <= 1 and 0 <= dw <= 1 else None)
                    if one is not None:
                        ll_shapes[j] = tuple(ll.shape[-2:])
                        ll, j, took_strip = one, j - 1, True
                        break
                res = ops.sfb2d_fused(ll, list(yh[j - n + 1:j + 1]), g0_row, g1_row, g0_col, g1_col, mode)
                if res is None:
                    n -= 1
            if took_strip:
                continue
            if res is not None:
                L = g0_row.numel()
                sh = tuple(ll.shape[-2:])
                for i in range(j, j - n, -1):
                    ll_shapes[i] = sh
                    sh = ((2 * yh[i].shape[-2], 2 * yh[i].shape[-1]) if mode == 2
                          else (2 * yh[i].shape[-2] - L + 2, 2 * yh[i].shape[-1] - L + 2))
                ll, j = res, j - n
                continue
            h = yh[j]
            ll_shapes[j] = tuple(ll.shape[-2:])
            if h is not None:
                if ll.shape[-2] > h.shape[-2]:
                    ll = ll[..., :-1, :]
                if ll.shape[-1] > h.shape[-1]:
                    ll = ll[..., :-1]
            ll = ops.sfb2d_best(ll, h, g0_row, g1_row, g0_col, g1_col, mode)
            j -= 1
        ctx.ll_shapes = ll_shapes
        return ll

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        with ops.hints(*ctx.hints):
            return SFB2DMulti._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        J = len(ctx.has_highs)
        grads = [None] * J
        d = None
        if any(ctx.needs_input_grad[:1]) or any(ctx.needs_input_grad[6 + j] for j in range(J)):
            g0_row, g1_row, g0_col, g1_col = ctx.saved_tensors
            d, j, L = dy, 0, g0_row.numel()
            while j < J:
                # Levels whose low-pass the forward handed on WHOLE (no 'unpad' between them: always so in periodization) are a plain
                # multi-level analysis with the synthesis taps: up to three of them in ONE launch of the fused kernel (round 6);
                # a level after which a dropped row / column gets its zero gradient back ends the group.
                n, res = 0, None
                h, w = d.shape[-2:]
                while FUSED_LEVELS and n < 3 and j + n < J and g0_col.numel() == L:
                    h, w = ops.coeff_len(h, L, ctx.mode), ops.coeff_len(w, L, ctx.mode)
                    n += 1
                    if (h, w) != tuple(ctx.ll_shapes[j + n - 1]):
                        break
                while n >= 2 and res is None:
                    res = ops.afb2d_fused(d, g0_row, g1_row, g0_col, g1_col, ctx.mode, n, whole=False)
                    if res is None:
                        n -= 1
                if res is None:
                    n = 1
                    d, dhigh = ops.afb2d_best(d, g0_row, g1_row, g0_col, g1_col, ctx.mode)
                    dhighs = [dhigh]
                else:
                    d, dhighs = res
                for i, dhigh in enumerate(dhighs):
                    if ctx.has_highs[j + i] and ctx.needs_input_grad[6 + j + i]:
                        grads[j + i] = dhigh
                j += n
                full = ctx.ll_shapes[j - 1]
                if tuple(d.shape[-2:]) != full:      # the forward dropped a row / column of this low-pass
                    d = torch.nn.functional.pad(d, (0, full[1] - d.shape[-1], 0, full[0] - d.shape[-2]))
            if not ctx.needs_input_grad[0]:
                d = None
        return (d, None, None, None, None, None) + tuple(grads)


_PAD_LL = False   # inner-level LL_j at a cache-line-aligned row pitch for the per-level path (measured neutral: off)


class AFB2DMulti(Function):
    """J analysis levels as ONE autograd node: ``AFB2DMulti.apply(x, h0_row, h1_row, h0_col, h1_col,
    mode_int, J) -> (yl, yh_0, ..., yh_{J-1})``.

    Forward: up to three levels at a time in ONE launch of the streaming kernel (wl_dwt2d_analysis_fused: LL_j stay
    in LDS) whenever the engine takes the configuration - enough planes to fill the chip, even tap count <= 12,
    16-byte rows - otherwise one specialised tile-kernel launch per level (generic kernel for unusual tap counts /
    float64).  Backward = the chain of J AFB2D.backward steps of the reference (synthesis with the stored analysis
    taps + crop, dwt/lowlevel.py:350-365), coarsest level first - which is an inverse transform with the analysis taps,
    so it runs on the streaming synthesis kernel too (the crops are its 'unpad')."""

    @staticmethod
    def forward(ctx, x, h0_row, h1_row, h0_col, h1_col, mode, J):
        _check_bank_mode(mode)
        ctx.save_for_backward(h0_row, h1_row, h0_col, h1_col)
        ctx.mode = mode
        ctx.hints = ops.current_hints()     # (see SFB2DMulti: the backward pass is an inverse transform with these taps - same hints)
        shapes, yh, ll, done = [], [], x, 0
        while done < J:
            n = min(4, J - done)
            # small planes (CNN feature maps, CIFAR-sized images): several planes per workgroup, up to four levels in LDS
            res = ops.afb2d_small(ll, h0_row, h1_row, h0_col, h1_col, mode, n) if FUSED_LEVELS else None
            if res is None:
                n = min(3, J - done)
            while FUSED_LEVELS and n >= 1 and res is None:   # e.g. periodization: one level per streaming launch
                res = ops.afb2d_fused(ll, h0_row, h1_row, h0_col, h1_col, mode, n, whole=J == 1)
                if res is None:
                    n -= 1
            if res is None:
                n = 1
                shapes.append(tuple(ll.shape[-2:]))
                ll, high = ops.afb2d_best(ll, h0_row, h1_row, h0_col, h1_col, mode, pad_ll=_PAD_LL and done + 1 < J, more_levels=done + 1 < J)
                yh.append(high)
            else:
                shapes.append(tuple(ll.shape[-2:]))
                ll, highs = res
                shapes.extend(tuple(h.shape[-2:]) for h in highs[:-1])
                yh.extend(highs)
            done += n
        ctx.shapes = shapes
        return (ll,) + tuple(yh)

    @staticmethod
    @once_differentiable
    def backward(ctx, dyl, *dyh):
        with ops.hints(*ctx.hints):
            return AFB2DMulti._backward(ctx, dyl, *dyh)

    @staticmethod
    def _backward(ctx, dyl, *dyh):
        dx = None
        if ctx.needs_input_grad[0]:
            h0_row, h1_row, h0_col, h1_col = ctx.saved_tensors
            dx, j = dyl, len(dyh) - 1
            while j >= 0:
                # the crop to the input size of each level is the 'unpad' of the inverse transform: up to three levels
                # in one launch of the streaming synthesis kernel, the last crop as a view
                n = min(4, j + 1)
                grp = list(dyh[j - n + 1:j + 1])
                res = (ops.sfb2d_small(dx, grp, h0_row, h1_row, h0_col, h1_col, ctx.mode)
                       if FUSED_LEVELS and all(g is not None for g in grp) else None)
                if res is None:
                    # the coarsest (j mod 3) + 1 levels first, so that the finest go three to a launch (see SFB2DMulti.forward)
                    n = min(3, j + 1, (j % 3) + 1 if j + 1 > 3 else 3)
                    if FUSED_LEVELS and n < min(4, j + 1):
                        grp = list(dyh[j - n + 1:j + 1])
                        if all(g is not None for g in grp):
                            res = ops.sfb2d_small(dx, grp, h0_row, h1_row, h0_col, h1_col, ctx.mode)
                took_strip = False
                while FUSED_LEVELS and n >= 1 and res is None:
                    grp = list(dyh[j - n + 1:j + 1])
                    ok = all(g is not None for g in grp)
                    if ok and n == 1 and 2 * grp[0].shape[-1] >= WIDE_ONE_LEVEL:
                        # a single wide level: the strip kernel first (SFB2DMulti.forward has the measurements)
                        one = ops.sfb2d_stream(dx, grp[0], h0_row, h1_row, h0_col, h1_col, ctx.mode, out_hw=ctx.shapes[j], force=ops.STREAM_FORCE)
                        if one is not None:
                            dx, j, took_strip = one, j - 1, True
                            break
                    res = ops.sfb2d_fused(dx, grp, h0_row, h1_row, h0_col, h1_col, ctx.mode) if ok else None
                    if res is None:
                        n -= 1
                if took_strip:
                    continue
                if res is not None:
                    j -= n
                    H, W = ctx.shapes[j + 1]
                    dx = res[..., :H, :W]
                    continue
                dx = ops.sfb2d_best(dx, dyh[j], h0_row, h1_row, h0_col, h1_col, ctx.mode, out_hw=ctx.shapes[j])
                j -= 1
            if not dx.is_contiguous():
                dx = dx.contiguous()
        return dx, None, None, None, None, None, None


class AFB1D(Function):
    """One level of 1-D analysis.  ``AFB1D.apply(x:(N,C,L), h0, h1, mode_int) -> (x0, x1)`` each (N,C,L')
    (reference dwt/lowlevel.py:368-424).  Backward = synthesis with the same stored taps, cropped to the input length."""

    @staticmethod
    def forward(ctx, x, h0, h1, mode):
        _check_bank_mode(mode)
        ctx.save_for_backward(h0, h1)
        ctx.shape = x.shape[2]
        ctx.mode = mode
        return ops.afb1d(x, h0, h1, mode, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dx0, dx1):
        dx = None
        if ctx.needs_input_grad[0]:
            h0, h1 = ctx.saved_tensors
            dx = ops.sfb1d(dx0, dx1, h0, h1, ctx.mode, 2, out_len=ctx.shape)
        return dx, None, None, None


class AFB1DMulti(Function):
    """All J levels of DWT1DForward as ONE autograd node: ``AFB1DMulti.apply(x, h0, h1, mode_int, J) -> (yl, yh_1 .. yh_J)`` =
    J x AFB1D chained (reference dwt/transform1d.py:44-59).  One launch of the fused 1-D kernel where the engine takes it
    (ops.afb1d_fused), the per-level launches otherwise; the backward is the chain of the per-level backward passes
    (synthesis with the same stored taps, cropped: reference dwt/lowlevel.py:409-424)."""

    @staticmethod
    def forward(ctx, x, h0, h1, mode, J):
        _check_bank_mode(mode)
        ctx.save_for_backward(h0, h1)
        ctx.mode = mode
        res = ops.afb1d_fused(x, h0, h1, mode, J) if FUSED_LEVELS else None
        if res is None:
            lo, his = x, []
            for _ in range(J):
                lo, hi = ops.afb1d(lo, h0, h1, mode, 2)
                his.append(hi)
        else:
            lo, his = res
        ctx.lens = [x.shape[2]] + [h.shape[2] for h in his[:-1]]
        return (lo,) + tuple(his)

    @staticmethod
    @once_differentiable
    def backward(ctx, dlo, *dhis):
        dx = None
        if ctx.needs_input_grad[0]:
            h0, h1 = ctx.saved_tensors
            dx = None
            if FUSED_LEVELS and len(dhis) <= 4 and all(d is not None for d in dhis):
                # one launch (csrc/wl_idwt1d_fused.h): the crops to the levels' input lengths are its 'unpad' / output length
                dx = ops.sfb1d_fused(dlo, list(dhis), h0, h1, ctx.mode, out_len=ctx.lens[0])
            if dx is None:
                dx = dlo
                for dh, n in zip(dhis[::-1], ctx.lens[::-1]):
                    dx = ops.sfb1d(dx, dh, h0, h1, ctx.mode, 2, out_len=n)
        return dx, None, None, None, None


class SFB1DMulti(Function):
    """All levels of DWT1DInverse as ONE autograd node: ``SFB1DMulti.apply(x0, g0, g1, mode_int, *highs) -> x`` with highs finest
    first, ``None`` entries = zeros (the level loop of DWT1DInverse.forward, reference dwt/transform1d.py:97-115, incl. the 'unpad'
    of a lowpass one sample longer than the next highpass).  One launch of the fused 1-D synthesis kernel where the engine takes
    it (ops.sfb1d_fused), the per-level launches otherwise; backward = the chain of SFB1D.backward steps (analysis with the stored
    synthesis taps, dwt/lowlevel.py:729-743), finest level first; a dropped sample gets a zero gradient."""

    @staticmethod
    def forward(ctx, x0, g0, g1, mode, *highs):
        _check_bank_mode(mode)
        ctx.save_for_backward(g0, g1)
        ctx.mode = mode
        ctx.has_highs = [h is not None for h in highs]
        J = len(highs)
        lo_lens = [None] * J            # length of the lowpass handed to level j, before the 'unpad'
        res = None
        if FUSED_LEVELS and 1 <= J <= 4 and all(ctx.has_highs):
            res = ops.sfb1d_fused(x0, list(highs), g0, g1, mode)
            if res is not None:
                L = g0.numel()
                n = x0.shape[-1]
                for j in range(J - 1, -1, -1):
                    lo_lens[j] = n
                    n = 2 * highs[j].shape[-1] - L + 2
        if res is None:
            res = x0
            for j in range(J - 1, -1, -1):
                x1 = highs[j]
                lo_lens[j] = res.shape[-1]
                if x1 is None:
                    x1 = torch.zeros_like(res)
                if res.shape[-1] > x1.shape[-1]:
                    res = res[..., :-1]
                res = ops.sfb1d(res, x1, g0, g1, mode, 2)
        ctx.lo_lens = lo_lens
        return res

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        J = len(ctx.has_highs)
        grads = [None] * J
        d = None
        if ctx.needs_input_grad[0] or any(ctx.needs_input_grad[4 + j] for j in range(J)):
            g0, g1 = ctx.saved_tensors
            d = dy
            for j in range(J):
                d, dh = ops.afb1d(d, g0, g1, ctx.mode, 2)
                if ctx.has_highs[j] and ctx.needs_input_grad[4 + j]:
                    grads[j] = dh
                if d.shape[-1] < ctx.lo_lens[j]:         # the forward dropped the last sample of this lowpass
                    d = torch.nn.functional.pad(d, (0, ctx.lo_lens[j] - d.shape[-1]))
            if not ctx.needs_input_grad[0]:
                d = None
        return (d, None, None, None) + tuple(grads)


class SFB1D(Function):
    """One level of 1-D synthesis.  ``SFB1D.apply(low, high, g0, g1, mode_int) -> y`` (reference dwt/lowlevel.py:697-743).
    Backward = analysis with the stored synthesis taps."""

    @staticmethod
    def forward(ctx, low, high, g0, g1, mode):
        _check_bank_mode(mode)
        ctx.mode = mode
        ctx.save_for_backward(g0, g1)
        return ops.sfb1d(low, high, g0, g1, mode, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dlow, dhigh = None, None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g0, g1 = ctx.saved_tensors
            dlow, dhigh = ops.afb1d(dy, g0, g1, ctx.mode, 2)
        return dlow, dhigh, None, None, None


def _as_taps(h, x):
    """Array-likes are the pywt-ordered filters (reversed into cross-correlation taps, as the reference does when it is
    handed arrays); tensors are taken as already prepared."""
    if isinstance(h, torch.Tensor):
        return h
    return torch.tensor(np.copy(np.array(h, dtype=np.float64).ravel()[::-1]), dtype=torch.float, device=x.device)


def afb1d(x, h0, h1, mode='zero', dim=-1):
    """Function-level 1-D analysis along one axis of a 4-D tensor (reference dwt/lowlevel.py:91-172): returns the
    lowpass and highpass sub-bands interleaved along the channel axis, (N, 2C, H', W') with channel 2c = low."""
    d = dim % 4
    lo, hi = ops.afb1d(x, _as_taps(h0, x), _as_taps(h1, x), mode_to_int(mode), d)
    n, c = lo.shape[:2]
    return torch.stack([lo, hi], dim=2).reshape(n, 2 * c, lo.shape[2], lo.shape[3])


def sfb1d(lo, hi, g0, g1, mode='zero', dim=-1):
    """Function-level 1-D synthesis along one axis of 4-D tensors (reference dwt/lowlevel.py:226-271); array-like
    filters are used as given (no reversal), like upstream."""
    d = dim % 4

    def prep(g):
        return g if isinstance(g, torch.Tensor) else torch.tensor(np.copy(np.array(g, dtype=np.float64).ravel()),
                                                                  dtype=torch.float, device=lo.device)
    return ops.sfb1d(lo, hi, prep(g0), prep(g1), mode_to_int(mode), d)


_ATROUS_EXT = {'zero': ops.EXT_ZERO, 'constant': ops.EXT_ZERO, 'symmetric': ops.EXT_SYM, 'reflect': ops.EXT_REFL,
               'periodic': ops.EXT_PERIODIC, 'replicate': ops.EXT_REPLICATE}


def afb1d_atrous(x, h0, h1, mode='periodic', dim=-1, dilation=1):
    """Undecimated (a-trous) 1-D analysis along one axis (reference dwt/lowlevel.py:175-223): the taps are dilated, the
    signal is padded by (L*dilation)//2 - dilation before and (L*dilation)//2 after with `mode`, the output keeps the
    input size.  Returns (N, 2C, H, W) with channel 2c = low.  NB like upstream the padding is done by ``mypad``, which
    knows 'symmetric', 'periodic', 'constant', 'reflect', 'replicate' and 'zero' - 'periodization' raises."""
    if mode not in _ATROUS_EXT:
        raise ValueError("Unkown pad type: {}".format(mode))
    d = dim % 4
    t0, t1 = _as_taps(h0, x), _as_taps(h1, x)
    L = t0.numel()
    L2 = (L * dilation) // 2
    n = x.shape[d]
    K = n + 2 * L2 - dilation - dilation * (L - 1)
    lo, hi = ops.corr1d(x, d, t0, t1, K, -(L2 - dilation), 1, dilation, _ATROUS_EXT[mode])
    nb, c = lo.shape[:2]
    return torch.stack([lo, hi], dim=2).reshape(nb, 2 * c, lo.shape[2], lo.shape[3])


def afb2d_atrous(x, filts, mode='periodization', dilation=1):
    """One undecimated 2-D level (reference dwt/lowlevel.py:475-521): rows then columns; returns (N, 4C, H, W) with
    channel 4c + 2r + b (r: band along W, b: band along H), i.e. (ll, lh, hl, hh) per input channel."""
    tensorize = [not isinstance(f, torch.Tensor) for f in filts]
    if len(filts) == 2:
        h0, h1 = filts
        if True in tensorize:
            h0_col, h1_col, h0_row, h1_row = prep_filt_afb2d(h0, h1, device=x.device)
        else:
            h0_col, h0_row, h1_col, h1_row = h0, h0.transpose(2, 3), h1, h1.transpose(2, 3)
    elif len(filts) == 4:
        if True in tensorize:
            h0_col, h1_col, h0_row, h1_row = prep_filt_afb2d(*filts, device=x.device)
        else:
            h0_col, h1_col, h0_row, h1_row = filts
    else:
        raise ValueError("Unknown form for input filts")
    if mode not in _ATROUS_EXT:
        raise ValueError("Unkown pad type: {}".format(mode))
    if FUSED_LEVELS and x.dim() == 4:
        # one launch per level (csrc/wl_swt2d.h): x read once, the four sub-bands written once in the returned layout
        y = ops.swt2d_level(x, _as_taps(h0_row, x), _as_taps(h1_row, x), _as_taps(h0_col, x), _as_taps(h1_col, x), dilation,
                            _ATROUS_EXT[mode])
        if y is not None:
            return y
    lohi = afb1d_atrous(x, h0_row, h1_row, mode=mode, dim=3, dilation=dilation)
    return afb1d_atrous(lohi, h0_col, h1_col, mode=mode, dim=2, dilation=dilation)


def prep_filt_afb2d_nonsep(h0_col, h1_col, h0_row=None, h1_row=None, device=None):
    """The four 2-D point-spread functions (ll, lh, hl, hh) of an analysis bank as a (4, 1, Ly, Lx) tensor, mirrored for
    cross-correlation (reference dwt/lowlevel.py:801-833)."""
    h0_col = np.array(h0_col).ravel()
    h1_col = np.array(h1_col).ravel()
    h0_row = h0_col if h0_row is None else np.array(h0_row).ravel()
    h1_row = h1_col if h1_row is None else np.array(h1_row).ravel()
    psf = [np.outer(c, r)[::-1, ::-1] for c, r in ((h0_col, h0_row), (h1_col, h0_row), (h0_col, h1_row), (h1_col, h1_row))]
    return torch.tensor(np.stack(psf)[:, None].copy(), dtype=torch.get_default_dtype(), device=device)


def prep_filt_sfb2d_nonsep(g0_col, g1_col, g0_row=None, g1_row=None, device=None):
    """The four 2-D point-spread functions of a synthesis bank as a (4, 1, Ly, Lx) tensor, not mirrored (reference
    dwt/lowlevel.py:836-867)."""
    g0_col = np.array(g0_col).ravel()
    g1_col = np.array(g1_col).ravel()
    g0_row = g0_col if g0_row is None else np.array(g0_row).ravel()
    g1_row = g1_col if g1_row is None else np.array(g1_row).ravel()
    psf = [np.outer(c, r) for c, r in ((g0_col, g0_row), (g1_col, g0_row), (g0_col, g1_row), (g1_col, g1_row))]
    return torch.tensor(np.stack(psf)[:, None], dtype=torch.get_default_dtype(), device=device)


class _AFB2DNonsep(Function):
    """afb2d_nonsep as an autograd node.  Backward = the true adjoint of the boundary gather + strided correlation in
    every mode - what autograd gives upstream, where the function is a plain ATen chain (dwt/lowlevel.py:524-597):
    mirrored / wrapped / repeated samples fold their gradient back onto their source (one kernel launch)."""

    @staticmethod
    def forward(ctx, x, filts, mode):
        ctx.save_for_backward(filts)
        ctx.mode, ctx.shape = mode, tuple(x.shape)
        return ops.afb2d_nonsep(x, filts, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dx = None
        if ctx.needs_input_grad[0]:
            filts, = ctx.saved_tensors
            dx = ops.afb2d_nonsep_bwd(dy, filts, ctx.mode, ctx.shape[-2:])
        return dx, None, None


class _SFB2DNonsep(Function):
    """sfb2d_nonsep as an autograd node.  Backward = an analysis of the gradient with the same point-spread functions
    (zero extension; periodic for periodization) - the exact adjoint in every mode, one kernel launch."""

    @staticmethod
    def forward(ctx, coeffs, filts, mode):
        ctx.save_for_backward(filts)
        ctx.mode, ctx.shape = mode, tuple(coeffs.shape)
        return ops.sfb2d_nonsep(coeffs, filts, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dc = None
        if ctx.needs_input_grad[0]:
            filts, = ctx.saved_tensors
            dc = ops.sfb2d_nonsep_bwd(dy, filts, ctx.mode, ctx.shape)
        return dc, None, None


def afb2d_nonsep(x, filts, mode='zero'):
    """One analysis level WITHOUT separate row and column filtering (reference dwt/lowlevel.py:524-597): ``filts`` =
    the (4,1,Ly,Lx) tensor of prep_filt_afb2d_nonsep, or a 2- / 4-tuple of 1-D banks.  Returns (N, 4C, H', W').
    'zero', 'symmetric', 'reflect', 'periodization' ('periodic' raises, as upstream)."""
    if isinstance(filts, (tuple, list)):
        filts = prep_filt_afb2d_nonsep(*filts, device=x.device)
    if mode not in ('zero', 'symmetric', 'reflect', 'periodization', 'per'):
        raise ValueError("Unkown pad type: {}".format(mode))
    return _AFB2DNonsep.apply(x, filts, mode_to_int(mode))


def sfb2d_nonsep(coeffs, filts, mode='zero'):
    """One synthesis level without separable filtering (reference dwt/lowlevel.py:746-798): coeffs (N,C,4,H,W) ->
    (N,C,2H-Ly+2,2W-Lx+2) (periodization: (N,C,2H,2W)); ``filts`` = the tensor of prep_filt_sfb2d_nonsep or a 2- /
    4-tuple of 1-D banks."""
    if isinstance(filts, (tuple, list)):
        if len(filts) not in (2, 4):
            raise ValueError("Unkown form for input filts")
        filts = prep_filt_sfb2d_nonsep(*filts, device=coeffs.device)
    if mode not in ('zero', 'symmetric', 'reflect', 'periodic', 'periodization', 'per'):
        raise ValueError("Unkown pad type: {}".format(mode))
    return _SFB2DNonsep.apply(coeffs, filts, mode_to_int(mode))


def afb2d(x, filts, mode='zero'):
    """Function-level analysis (reference dwt/lowlevel.py:427-472): ``filts`` is a 2- or 4-tuple
    of arrays / tensors (h0_col, h1_col[, h0_row, h1_row]); here the *col* pair really filters
    along H (the Module path swaps them, quirk Q1).  Returns (N, 4C, H', W')."""
    tensorize = [not isinstance(f, torch.Tensor) for f in filts]
    if len(filts) == 2:
        h0, h1 = filts
        if True in tensorize:
            h0_col, h1_col, h0_row, h1_row = prep_filt_afb2d(h0, h1, device=x.device)
        else:
            h0_col, h0_row, h1_col, h1_row = h0, h0.transpose(2, 3), h1, h1.transpose(2, 3)
    elif len(filts) == 4:
        if True in tensorize:
            h0_col, h1_col, h0_row, h1_row = prep_filt_afb2d(*filts, device=x.device)
        else:
            h0_col, h1_col, h0_row, h1_row = filts
    else:
        raise ValueError("Unknown form for input filts")
    low, highs = AFB2D.apply(x, h0_row, h1_row, h0_col, h1_col, mode_to_int(mode))
    n, c = low.shape[:2]
    return torch.cat([low[:, :, None], highs], dim=2).reshape(n, 4 * c, low.shape[-2], low.shape[-1])


def sfb2d(ll, lh, hl, hh, filts, mode='zero'):
    """Function-level synthesis (reference dwt/lowlevel.py:600-644)."""
    tensorize = [not isinstance(f, torch.Tensor) for f in filts]
    if len(filts) == 2:
        g0, g1 = filts
        if True in tensorize:
            g0_col, g1_col, g0_row, g1_row = prep_filt_sfb2d(g0, g1, device=ll.device)
        else:
            g0_col, g0_row, g1_col, g1_row = g0, g0.transpose(2, 3), g1, g1.transpose(2, 3)
    elif len(filts) == 4:
        if True in tensorize:
            g0_col, g1_col, g0_row, g1_row = prep_filt_sfb2d(*filts, device=ll.device)
        else:
            g0_col, g1_col, g0_row, g1_row = filts
    else:
        raise ValueError("Unknown form for input filts")
    highs = torch.stack([lh, hl, hh], dim=2)
    return SFB2D.apply(ll, highs, g0_row, g1_row, g0_col, g1_col, mode_to_int(mode))


# ---- filter preparation (buffer shapes/orders are part of the state_dict contract) -----------------
def _vec(h, reverse, device):
    h = np.array(h, dtype=np.float64).ravel()
    if reverse:
        h = h[::-1].copy()
    return torch.tensor(h, device=device, dtype=torch.get_default_dtype())


def prep_filt_afb1d(h0, h1, device=None):
    """Analysis taps are stored reversed, shape (1,1,L) (reference dwt/lowlevel.py:956-975)."""
    return _vec(h0, True, device).reshape(1, 1, -1), _vec(h1, True, device).reshape(1, 1, -1)


def prep_filt_sfb1d(g0, g1, device=None):
    """Synthesis taps are stored as given (reference dwt/lowlevel.py:902-922)."""
    return _vec(g0, False, device).reshape(1, 1, -1), _vec(g1, False, device).reshape(1, 1, -1)


def _to_2d(col0, col1, row0, row1):
    return (col0.reshape(1, 1, -1, 1), col1.reshape(1, 1, -1, 1),
            row0.reshape(1, 1, 1, -1), row1.reshape(1, 1, 1, -1))


def prep_filt_afb2d(h0_col, h1_col, h0_row=None, h1_row=None, device=None):
    """(h0_col, h1_col, h0_row, h1_row) with shapes (1,1,L,1) / (1,1,1,L)
    (reference dwt/lowlevel.py:925-953)."""
    c0, c1 = prep_filt_afb1d(h0_col, h1_col, device)
    r0, r1 = (c0, c1) if h0_row is None else prep_filt_afb1d(h0_row, h1_row, device)
    return _to_2d(c0, c1, r0, r1)


def prep_filt_sfb2d(g0_col, g1_col, g0_row=None, g1_row=None, device=None):
    """(g0_col, g1_col, g0_row, g1_row) (reference dwt/lowlevel.py:870-899)."""
    c0, c1 = prep_filt_sfb1d(g0_col, g1_col, device)
    r0, r1 = (c0, c1) if g0_row is None else prep_filt_sfb1d(g0_row, g1_row, device)
    return _to_2d(c0, c1, r0, r1)
