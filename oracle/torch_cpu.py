"""PyTorch-CPU restatement of the reference's conv formulation of the 2-D DWT (TEST / BASELINE INFRASTRUCTURE ONLY).

The reference computes every level as index-gather + grouped strided ``conv2d`` (analysis, dwt/lowlevel.py:91-172) and
grouped ``conv_transpose2d`` (synthesis, :226-271) on ATen.  /root/reference cannot travel to the GPU box, so this file
restates that formulation - the same ATen operators, the same operator count per level, multi-threaded through torch's
intra-op pool - for ``bench.py``'s ``cpu_baseline`` ("kind": "restated-torch").  It is pinned to the reference's golden
vectors by tests/test_oracle_golden.py and is never imported by the product package.

    analysis  (one axis): xe = x[..., ext(arange(2K+L-2) + base)]   (gather: the reference's mypad / roll)
                          lo, hi = conv2d(xe, stack(h0,h1) per channel, stride 2, groups=C)
    synthesis (one axis): full = conv_transpose2d(lo, g0, stride 2, groups=C) + conv_transpose2d(hi, g1, ...)
                          crop (non-periodization) or fold + roll (periodization)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import wavelet_oracle as wo


def _ext_index(n, L, mode, K):
    base = (1 - L // 2) if mode in ('per', 'periodization') else -((2 * (K - 1) - n + L) // 2)
    e = np.arange(2 * K + L - 2) + base
    if mode in ('per', 'periodization'):
        ne = n + (n & 1)
        j = np.mod(e, ne)
        return np.where(j == n, n - 1, j), None
    src, valid = wo.ext_index(e, n, mode)
    return src, (None if valid.all() else valid)


def _afb1d(x, h0, h1, mode, dim):
    """x (N,C,H,W); filter along dim (2 or 3) -> (lo, hi).  One gather + one grouped strided conv (2C outputs)."""
    N, C = x.shape[:2]
    n, L = x.shape[dim], h0.numel()
    K = wo.dwt_coeff_len(n, L, mode)
    if mode in ('per', 'periodization') and n + (n & 1) < L - 1:
        raise NotImplementedError('single-fold periodization quirk is not restated here')
    src, valid = _ext_index(n, L, mode, K)
    xe = x.index_select(dim, torch.as_tensor(src))
    if valid is not None:
        shape = [1, 1, 1, 1]
        shape[dim] = -1
        xe = xe * torch.as_tensor(valid, dtype=x.dtype).reshape(shape)
    w = torch.stack([h0.reshape(-1), h1.reshape(-1)]).to(x.dtype)            # (2, L); conv2d cross-correlates
    w = w.reshape(2, 1, L, 1) if dim == 2 else w.reshape(2, 1, 1, L)
    y = F.conv2d(xe, w.repeat(C, 1, 1, 1), stride=(2, 1) if dim == 2 else (1, 2), groups=C)
    y = y.reshape(N, C, 2, y.shape[-2], y.shape[-1])
    return y[:, :, 0], y[:, :, 1]


def _sfb1d(lo, hi, g0, g1, mode, dim):
    N, C = lo.shape[:2]
    L, K = g0.numel(), lo.shape[dim]
    shp = (1, 1, L, 1) if dim == 2 else (1, 1, 1, L)
    st = (2, 1) if dim == 2 else (1, 2)
    # conv_transpose2d: full[n] = sum_k lo[k] g[n - 2k]
    full = (F.conv_transpose2d(lo, g0.reshape(shp).to(lo.dtype).repeat(C, 1, 1, 1), stride=st, groups=C) +
            F.conv_transpose2d(hi, g1.reshape(shp).to(lo.dtype).repeat(C, 1, 1, 1), stride=st, groups=C))
    if mode in ('per', 'periodization'):
        n = 2 * K
        head = full.narrow(dim, 0, n).clone()
        if L > 2:
            head.narrow(dim, 0, L - 2).add_(full.narrow(dim, n, L - 2))
        return torch.roll(head, 1 - L // 2, dims=dim)
    return full.narrow(dim, L - 2, 2 * K - L + 2)


def dwt_forward(x, J, h0, h1, mode):
    """DWTForward (dwt/transform2d.py:63-74) with one filter pair for both axes; taps = the stored (reversed) ones."""
    h0, h1 = torch.as_tensor(np.asarray(h0)), torch.as_tensor(np.asarray(h1))
    yh, ll = [], x
    for _ in range(J):
        lo, hi = _afb1d(ll, h0, h1, mode, 3)
        ll, lh = _afb1d(lo, h0, h1, mode, 2)
        hl, hh = _afb1d(hi, h0, h1, mode, 2)
        yh.append(torch.stack([lh, hl, hh], dim=2))
    return ll, yh


def dwt_inverse(yl, yh, g0, g1, mode):
    """DWTInverse (dwt/transform2d.py:131-148)."""
    g0, g1 = torch.as_tensor(np.asarray(g0)), torch.as_tensor(np.asarray(g1))
    ll = yl
    for h in yh[::-1]:
        if ll.shape[-2] > h.shape[-2]:
            ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]:
            ll = ll[..., :-1]
        lh, hl, hh = h.unbind(2)
        lo = _sfb1d(ll, lh, g0, g1, mode, 2)
        hi = _sfb1d(hl, hh, g0, g1, mode, 2)
        ll = _sfb1d(lo, hi, g0, g1, mode, 3)
    return ll
