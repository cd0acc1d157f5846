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
