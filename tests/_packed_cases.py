"""Several PLANES per workgroup on narrow levels (csrc/wl_dwt_strip.h `run`, wl_idwt_strip.h): when a level's whole row is one
strip that keeps only two / one of a workgroup's four compute waves busy (256 / 128 output columns: the deeper levels of a wide
pyramid), the one-level strip kernels give a workgroup 2 / 4 planes - each with its own staged ring and compute waves, every
stager wave taking its row of each plane.  Every case against the ORACLE on a sample of planes (first, last, around the boundaries
of the plane groups, the short last group) and, when `packed` is asked for, with the launch's grid as the witness that the planes
were packed.  Shared by the emulator tests (device 'cpu' under emu_backend.emulated(): a 2-CU chip) and the -m gpu tests."""
import numpy as np
import torch

import pytorch_wavelets_amd as pw
from oracle import wavelet_oracle as wo


def _flat(b):
    return b.detach().cpu().double().numpy().ravel()


def _rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# (wave, mode, dtype, H, W): W / 2 output columns decide the planes per workgroup (<= 128: four, <= 256: two)
PACKED_CASES = [
    ('db4', 'symmetric', torch.float32, 40, 100),        # four planes, odd band width (element stores)
    ('db8', 'periodization', torch.float16, 48, 256),    # config 5's level 4 in small: lattice variant, four planes
    ('db8', 'periodization', torch.float16, 32, 512),    # its level 3: two planes
    ('db6', 'zero', torch.float32, 36, 300),             # two planes, zero padding (rows of zeros staged for every plane)
    ('db2', 'reflect', torch.float32, 34, 130),          # four planes, mirrored halo cells
    ('db10', 'periodic', torch.float32, 40, 448),        # two planes, wrapped pieces, lattice variant
    ('sym7', 'symmetric', torch.float16, 30, 200),       # 14 taps
]


def check_packed(dev, wave, mode, dtype, H, W, planes, packed=True):
    """One analysis level and its synthesis on the forced strip kernels, `planes` planes (not a multiple of four)."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(41)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS, ops.LATTICE_MIN_ELEMS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS, ops.LATTICE_MIN_ELEMS = True, False, 0
    tol = 1e-5 if dtype == torch.float32 else 3e-3
    try:
        x = torch.tensor(rng.randn(planes, 1, H, W), dtype=dtype, device=dev)
        fwd = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev).to(dtype)
        inv = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dtype)
        be = ops._backend()
        yl, yh = fwd(x)
        assert 'WlAfbStrip<' in pw.last_kernel(), pw.last_kernel()
        g_fwd = int(be.wl_last_grid())
        y = inv((yl, yh))
        assert 'WlSfbStrip<' in pw.last_kernel(), pw.last_kernel()
        g_inv = int(be.wl_last_grid())
        if packed:      # fewer workgroups than planes: only a launch that packs planes gets there
            assert g_fwd < planes and g_inv < planes, (wave, mode, g_fwd, g_inv, planes)
        sample = sorted(set([0, 1, 3, 4, 5, 7, 8, planes // 2, planes - 6, planes - 5, planes - 4, planes - 3, planes - 2, planes - 1]))
        sample = [p for p in sample if 0 <= p < planes]
        idx = torch.tensor(sample, device=dev)
        xs = x[idx].detach().cpu().double().numpy()
        oyl, oyh = wo.dwt_forward(xs, 1, _flat(fwd.h0_col), _flat(fwd.h1_col), _flat(fwd.h0_row), _flat(fwd.h1_row), mode)
        e = max(_rel(yl[idx], oyl), _rel(yh[0][idx], oyh[0]))
        assert e <= tol, ('forward', wave, mode, e)
        # the synthesis against the oracle ON THE ENGINE'S OWN COEFFICIENTS (as stored: float16 rounding is then the input's, not an error)
        cl = yl[idx].detach().cpu().double().numpy()
        ch = yh[0][idx].detach().cpu().double().numpy()
        oy = wo.dwt_inverse(cl, [ch], _flat(inv.g0_col), _flat(inv.g1_col), _flat(inv.g0_row), _flat(inv.g1_row), mode)
        e2 = _rel(y[idx], oy)
        assert e2 <= tol, ('inverse', wave, mode, e2)
        return e, e2
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS, ops.LATTICE_MIN_ELEMS = prev


# ---- the fused multi-level analysis on a ROW-PADDED input (wl_dwt2d_analysis_fused_ex) ------------------------------------
PADDED_FUSED_CASES = [
    # (wave, mode, H, W, nlev): W * 4 no multiple of 16 - the row ends inside its last 16-byte piece
    ('db4', 'symmetric', 75, 515, 2),      # the ll below a 1024-wide image: three 1 KiB pieces per row
    ('db2', 'zero', 40, 515, 2),           # zero mode: the cells behind the row must be cleared, not left to the (NaN) padding
    ('db3', 'reflect', 44, 261, 3),
    ('haar', 'zero', 36, 130, 2),
    ('db4', 'zero', 52, 259, 1),
    ('db6', 'symmetric', 48, 323, 2),
]


def check_padded_fused(dev, wave, mode, H, W, nlev, planes=6, dtype=torch.float32):
    """afb2d_fused on a view of a buffer whose row pitch is the next whole number of 16-byte pieces and whose padding holds NaN:
    equal to the oracle, no NaN anywhere (what lies behind a row is loaded with its last piece and must never be read)."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(43)
    q = 16 // torch.tensor([], dtype=dtype).element_size()
    pitch = (W + q - 1) // q * q
    assert pitch != W
    buf = torch.full((planes, 1, H, pitch), float('nan'), dtype=dtype, device=dev)
    x = torch.tensor(rng.randn(planes, 1, H, W), dtype=dtype, device=dev)
    buf[..., :W] = x
    xfm = pw.DWTForward(J=nlev, wave=wave, mode=mode).to(dev).to(dtype)
    imode = {'zero': 0, 'symmetric': 1, 'reflect': 4}[mode]
    prev = ops.FUSED_STRIPS
    ops.FUSED_STRIPS = 1
    try:
        res = ops.afb2d_fused(buf[..., :W], xfm.h0_col, xfm.h1_col, xfm.h0_row, xfm.h1_row, imode, nlev)
    finally:
        ops.FUSED_STRIPS = prev
    assert res is not None, 'the fused kernel declined a row-padded input'
    assert 'WlAfbRows<' in pw.last_kernel()
    yl, yh = res
    oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), nlev, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row), _flat(xfm.h1_row), mode)
    assert bool(torch.isfinite(yl).all()) and all(bool(torch.isfinite(h).all()) for h in yh)
    e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
    assert e <= (1e-5 if dtype == torch.float32 else 3e-3), (wave, mode, e)
    return e


def check_wide_pyramid(dev, wave='db4', mode='symmetric', shape=(2, 3, 96, 1024), J=3):
    """A 1024-wide pyramid through the module: level 1 on the strip kernel (its ll, 515 columns, written at a padded pitch), the
    remaining levels in ONE launch of the fused kernel - against the oracle, forward and gradient-free inverse round trip."""
    rng = np.random.RandomState(47)
    x = torch.tensor(rng.randn(*shape), dtype=torch.float32, device=dev)
    xfm = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev)
    ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev)
    c0 = pw.launch_count()
    yl, yh = xfm(x)
    ks = [k for k in pw.kernels_since(c0) if 'armed' not in k and 'aux' not in k]
    assert len(ks) == 2 and ks[0].startswith('WlAfbStrip<') and ks[1].startswith('WlAfbRows<'), ks
    assert yl.is_contiguous() and all(h.is_contiguous() for h in yh)
    oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), J, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row), _flat(xfm.h1_row), mode)
    e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
    assert e <= 1e-5, e
    rec = ifm((yl, yh))
    assert float((rec - x).abs().max()) <= 1e-4 * float(x.abs().max())
    return e
