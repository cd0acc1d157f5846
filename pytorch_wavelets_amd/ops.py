"""Tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw device pointers and
the current HIP stream down, nothing else.  (PyTorch is plumbing here: memory + streams.)"""
import torch

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}
_TEST_BACKEND = None   # set ONLY by tests/emu_backend.py (host emulation of the kernel sources)


def _backend():
    return _TEST_BACKEND if _TEST_BACKEND is not None else _lib.get()


def _check_tensor(t, name):
    if t.dtype not in _DTYPES:
        raise TypeError('%s: unsupported dtype %s (float16/float32/float64)' % (name, t.dtype))
    if not t.is_cuda and _TEST_BACKEND is None:
        raise RuntimeError('%s is on %s: pytorch_wavelets_amd runs on MI355X only (HIP kernels, '
                           'no CPU fallback). Move the module and its input to a cuda device.'
                           % (name, t.device))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


def _taps(h, ref):
    """1-D contiguous taps on ref's device in the accumulate dtype (float, or double for f64)."""
    acc = torch.float64 if ref.dtype == torch.float64 else torch.float32
    return h.detach().reshape(-1).to(device=ref.device, dtype=acc).contiguous()


def coeff_len(n, L, mode):
    return (n + 1) // 2 if mode == 2 else (n + L - 1) // 2


def afb2d(x, h_w_lo, h_w_hi, h_h_lo, h_h_hi, mode):
    """One analysis level: x (N,C,H,W) -> ll (N,C,Kh,Kw), highs (N,C,3,Kh,Kw)."""
    _check_tensor(x, 'x')
    x = x.contiguous()
    N, C, H, W = x.shape
    hwl, hwh, hhl, hhh = (_taps(h, x) for h in (h_w_lo, h_w_hi, h_h_lo, h_h_hi))
    Lw, Lh = hwl.numel(), hhl.numel()
    Kh, Kw = coeff_len(H, Lh, mode), coeff_len(W, Lw, mode)
    ll = torch.empty((N, C, Kh, Kw), dtype=x.dtype, device=x.device)
    highs = torch.empty((N, C, 3, Kh, Kw), dtype=x.dtype, device=x.device)
    rc = _backend().wl_dwt2d_analysis(x.data_ptr(), ll.data_ptr(), highs.data_ptr(), _DTYPES[x.dtype],
                                      N * C, H, W, hwl.data_ptr(), hwh.data_ptr(), Lw, hhl.data_ptr(),
                                      hhh.data_ptr(), Lh, mode, _stream(x))
    _lib.check(rc, 'wl_dwt2d_analysis')
    return ll, highs


def sfb2d(ll, highs, g_w_lo, g_w_hi, g_h_lo, g_h_hi, mode, out_hw=None):
    """One synthesis level: ll (N,C,Kh,Kw) [may be a strided crop], highs (N,C,3,Kh,Kw) or None
    -> y (N,C,OH,OW); out_hw crops the result (used by the analysis backward)."""
    _check_tensor(ll, 'll')
    N, C, Kh, Kw = ll.shape
    if ll.stride(3) != 1 or ll.stride(0) != C * ll.stride(1) or ll.numel() == 0:
        ll = ll.contiguous()
    if highs is not None:
        if highs.dtype != ll.dtype:
            highs = highs.to(ll.dtype)
        highs = highs.contiguous()
        assert highs.shape == (N, C, 3, Kh, Kw), (highs.shape, ll.shape)
    gwl, gwh, ghl, ghh = (_taps(g, ll) for g in (g_w_lo, g_w_hi, g_h_lo, g_h_hi))
    Lw, Lh = gwl.numel(), ghl.numel()
    OH = 2 * Kh if mode == 2 else 2 * Kh - Lh + 2
    OW = 2 * Kw if mode == 2 else 2 * Kw - Lw + 2
    if out_hw is not None:
        OH, OW = min(OH, out_hw[0]), min(OW, out_hw[1])
    y = torch.empty((N, C, OH, OW), dtype=ll.dtype, device=ll.device)
    rc = _backend().wl_dwt2d_synthesis(ll.data_ptr(), ll.stride(1), ll.stride(2),
                                       None if highs is None else highs.data_ptr(), y.data_ptr(),
                                       _DTYPES[ll.dtype], N * C, Kh, Kw, OH, OW, gwl.data_ptr(),
                                       gwh.data_ptr(), Lw, ghl.data_ptr(), ghh.data_ptr(), Lh, mode,
                                       _stream(ll))
    _lib.check(rc, 'wl_dwt2d_synthesis')
    return y


def afb2d_fused(x, h_w_lo, h_w_hi, h_h_lo, h_h_hi, mode, nlev, strips=0):
    """`nlev` analysis levels in one launch (LL_j stay in LDS).  Returns (yl, [yh_0..]) or None when
    the streaming kernel does not cover the configuration (caller goes level by level)."""
    import ctypes
    import os
    _check_tensor(x, 'x')
    if x.dtype == torch.float64 or nlev < 1 or nlev > 4 or os.environ.get('WL_DISABLE_FUSED'):
        return None
    x = x.contiguous()
    N, C, H, W = x.shape
    hwl, hwh, hhl, hhh = (_taps(h, x) for h in (h_w_lo, h_w_hi, h_h_lo, h_h_hi))
    L = hwl.numel()
    if hhl.numel() != L:
        return None
    yh = []
    h, w = H, W
    for _ in range(nlev):
        h, w = coeff_len(h, L, mode), coeff_len(w, L, mode)
        yh.append(torch.empty((N, C, 3, h, w), dtype=x.dtype, device=x.device))
    yl = torch.empty((N, C, h, w), dtype=x.dtype, device=x.device)
    ptrs = (ctypes.c_void_p * nlev)(*[t.data_ptr() for t in yh])
    rc = _backend().wl_dwt2d_analysis_fused(x.data_ptr(), yl.data_ptr(), ptrs, _DTYPES[x.dtype], N * C, H, W,
                                            nlev, hwl.data_ptr(), hwh.data_ptr(), hhl.data_ptr(),
                                            hhh.data_ptr(), L, mode, strips, _stream(x))
    if rc == -3:
        return None
    _lib.check(rc, 'wl_dwt2d_analysis_fused')
    return yl, yh
