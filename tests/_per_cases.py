"""Round 6: several PERIODIZATION levels in one launch of the fused streaming analysis kernel (csrc/wl_dwt_rows.h: the rows above /
below a level's plane are computed by the level above - WlRowsSched::src_row - and tap counts with L % 4 == 0, whose samples sit
on odd cells, run the ODD instantiations).  Every case against the ORACLE (reference dwt/lowlevel.py:134-150 through
oracle/wavelet_oracle.py).  Shared by the emulator tests (device 'cpu' under emu_backend.emulated()) and the -m gpu tests."""
import numpy as np
import torch

import pytorch_wavelets_amd as pw
from oracle import wavelet_oracle as wo


def _flat(b):
    return b.detach().cpu().double().numpy().ravel()


def _rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# (wave, H, W, J, dtype, strips): strips 1 = whole planes, 2 = every plane cut in two, 0 = the launcher's own policy
FUSED_PER_CASES = [
    ('haar', 64, 64, 3, torch.float32, 1),
    ('db2', 96, 128, 3, torch.float32, 2),          # L % 4 == 0: the odd-cell instantiations
    ('db3', 64, 96, 2, torch.float32, 1),
    ('db4', 128, 256, 3, torch.float32, 1),
    ('db4', 128, 256, 3, torch.float32, 2),
    ('db4', 80, 272, 2, torch.float32, 0),          # two 1 KiB pieces per row
    ('db4', 72, 520, 2, torch.float32, 1),          # three pieces per row
    ('db5', 96, 128, 3, torch.float32, 2),
    ('db6', 96, 128, 3, torch.float32, 1),          # 12 taps, odd cells: lattice variant
    ('db7', 128, 128, 2, torch.float32, 2),
    ('db8', 128, 256, 3, torch.float32, 1),
    ('db7', 128, 512, 3, torch.float16, 1),         # 2 KiB rows of float16 (12 / 16 / 20 taps in float16 periodization stay on the strip kernels: policy)
    ('db5', 128, 256, 3, torch.float16, 2),
    ('db10', 160, 160, 2, torch.float32, 1),
    ('sym4', 64, 192, 3, torch.float16, 0),
]


def check_fused_periodization(dev, wave, H, W, J, dtype, strips, planes=(2, 2), require_fused=True):
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(71)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = strips, 0
    try:
        x = torch.tensor(rng.randn(planes[0], planes[1], H, W), dtype=dtype, device=dev)
        xfm = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev).to(dtype)
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        ks = [k for k in pw.kernels_since(c0) if not k.endswith(')')]
        if require_fused:   # ONE launch that does the transform's work
            assert len(ks) == 1 and ks[0].startswith('WlAfbRows<'), (wave, H, W, J, ks)
        oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), J, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row),
                                  _flat(xfm.h1_row), 'periodization')
        e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
        assert e <= (1e-5 if dtype == torch.float32 else 4e-3), (wave, H, W, J, e, ks)
        return e
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


def check_fused_periodization_corners(dev):
    """What the fused kernel must DECLINE in periodization (the per-level kernels restate these corners): a level shorter than the
    filter (the reference's wrap-add folds once: no periodic convolution), an odd number of rows or columns below level 1 (the
    repeated last row / column) - and the module's answer is the oracle's either way."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(73)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = 1, 0
    try:
        for wave, H, W, J in (('db10', 64, 64, 3), ('db4', 68, 64, 3), ('db8', 36, 260, 2), ('db6', 50, 50, 2), ('db4', 64, 132, 3)):
            x = torch.tensor(rng.randn(1, 2, H, W), dtype=torch.float32, device=dev)
            xfm = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev)
            yl, yh = xfm(x)
            oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), J, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row),
                                      _flat(xfm.h1_row), 'periodization')
            e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
            assert e <= 1e-5, (wave, H, W, J, e)
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


def check_periodization_gradient(dev, shape=(2, 2, 64, 128), wave='db4', J=3):
    """The module's backward pass (an inverse transform with the analysis taps) next to the fused periodization forward: the
    gradient of sum(w . coefficients) is the adjoint applied to w - compared with the oracle's forward by the inner-product test."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(79)
    prev = ops.FUSED_STRIPS
    ops.FUSED_STRIPS = 1
    try:
        x = torch.tensor(rng.randn(*shape), dtype=torch.float32, device=dev, requires_grad=True)
        xfm = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev)
        yl, yh = xfm(x)
        wl = torch.tensor(rng.randn(*yl.shape), dtype=torch.float32, device=dev)
        wh = [torch.tensor(rng.randn(*h.shape), dtype=torch.float32, device=dev) for h in yh]
        loss = (yl * wl).sum() + sum((h * w).sum() for h, w in zip(yh, wh))
        g, = torch.autograd.grad(loss, x)
        # <A x', w> == <x', A^T w> for a random x' (A from the oracle)
        xp = rng.randn(*shape)
        oyl, oyh = wo.dwt_forward(xp, J, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row), _flat(xfm.h1_row), 'periodization')
        lhs = float((oyl * wl.cpu().double().numpy()).sum() + sum((a * w.cpu().double().numpy()).sum() for a, w in zip(oyh, wh)))
        rhs = float((xp * g.detach().cpu().double().numpy()).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    finally:
        ops.FUSED_STRIPS = prev


def check_inverse_backward_is_one_fused_analysis(dev, shape=(2, 2, 64, 128), wave='db4', J=3, mode='periodization'):
    """DWTInverse.backward = J analysis levels with the synthesis taps (reference dwt/lowlevel.py:683-694 chained by autograd); where the
    forward 'unpadded' nothing between levels - always in periodization - they run as ONE fused launch.  Adjoint test against the oracle's
    inverse: <S c', w> == <c', S^T w>."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(83)
    prev = ops.FUSED_STRIPS
    ops.FUSED_STRIPS = 1
    try:
        xfm = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev)
        ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev)
        with torch.no_grad():
            yl, yh = xfm(torch.tensor(rng.randn(*shape), dtype=torch.float32, device=dev))
        yl = yl.clone().requires_grad_(True)
        yh = [h.clone().requires_grad_(True) for h in yh]
        y = ifm((yl, yh))
        w = torch.tensor(rng.randn(*y.shape), dtype=torch.float32, device=dev)
        c0 = pw.launch_count()
        grads = torch.autograd.grad((y * w).sum(), [yl] + yh)
        ks = [k for k in pw.kernels_since(c0) if not k.endswith(')') and k.startswith('Wl')]
        assert len(ks) == 1 and ks[0].startswith('WlAfbRows<'), ks
        cl = rng.randn(*yl.shape)
        ch = [rng.randn(*h.shape) for h in yh]
        oy = wo.dwt_inverse(cl, ch, _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), mode)
        lhs = float((oy * w.cpu().double().numpy()).sum())
        rhs = float((cl * grads[0].cpu().double().numpy()).sum() + sum((a * g.cpu().double().numpy()).sum() for a, g in zip(ch, grads[1:])))
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    finally:
        ops.FUSED_STRIPS = prev


# ---- round 6: the fused SYNTHESIS in periodization (csrc/wl_idwt_rows.h, PER = 1) ----------------------------------------------------------
# (wave, H, W, J, dtype, strips)
FUSED_IPER_CASES = [
    ('haar', 64, 64, 3, torch.float32, 1),
    ('db2', 96, 128, 3, torch.float32, 2),           # odd L/2 - 1: the roll splits one output pair between a row's last and first column
    ('db3', 64, 96, 2, torch.float32, 1),
    ('db4', 128, 256, 3, torch.float32, 1),
    ('db4', 128, 256, 3, torch.float32, 2),          # both halves of a cut plane: rotated frames
    ('db4', 80, 336, 2, torch.float32, 0),           # (the policy takes planes of 320 columns and more)
    ('db4', 72, 520, 1, torch.float32, 1),           # rows of more than one 1 KiB DMA piece + an 8-byte tail
    ('db5', 96, 128, 3, torch.float32, 2),
    ('db6', 96, 128, 3, torch.float32, 1),
    ('db8', 128, 256, 3, torch.float32, 1),          # lattice variant
    ('db9', 160, 160, 2, torch.float32, 2),
    ('db10', 160, 160, 2, torch.float32, 1),
    ('bior2.2', 64, 128, 3, torch.float32, 1),       # biorthogonal: the four-bank direct form
    ('db4', 128, 256, 3, torch.float16, 2),
    ('db8', 128, 512, 3, torch.float16, 1),
]


def check_fused_periodization_inverse(dev, wave, H, W, J, dtype, strips, planes=(2, 2), require_fused=True):
    """DWTInverse in periodization on the fused synthesis kernel (all levels in ONE launch) against the oracle
    (reference dwt/lowlevel.py:252-261 through oracle/wavelet_oracle.py) on random coefficients."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(89)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops.IROWS_F16_MAXL
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops.IROWS_F16_MAXL = strips, 0, 0, 99
    try:
        ifm = pw.DWTInverse(wave=wave, mode='periodization').to(dev).to(dtype)
        h, w, shp = H, W, []
        for _ in range(J):
            h, w = h // 2, w // 2
            shp.append((h, w))
        yl = torch.tensor(rng.randn(planes[0], planes[1], h, w), dtype=dtype, device=dev)
        yh = [torch.tensor(rng.randn(planes[0], planes[1], 3, a, b), dtype=dtype, device=dev) for a, b in shp]
        c0 = pw.launch_count()
        y = ifm((yl, yh))
        ks = [k for k in pw.kernels_since(c0) if not k.endswith(')')]
        if require_fused:
            assert len(ks) == 1 and ks[0].startswith('WlSfbRows<'), (wave, H, W, J, ks)
        oy = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [t.detach().cpu().double().numpy() for t in yh], _flat(ifm.g0_col), _flat(ifm.g1_col),
                            _flat(ifm.g0_row), _flat(ifm.g1_row), 'periodization')
        e = _rel(y, oy)
        assert tuple(y.shape[-2:]) == (H, W) and e <= (1e-5 if dtype == torch.float32 else 4e-3), (wave, H, W, J, e, ks)
        return e
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops.IROWS_F16_MAXL = prev


def check_fused_periodization_inverse_corners(dev):
    """Shapes the fused periodized synthesis must decline (the per-level kernels restate them) - an odd level (the module's 'unpad'), a level
    shorter than the filter, None for a band - and the round trip x -> DWTForward -> DWTInverse -> x through both fused kernels."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(97)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML = 1, 0, 0
    try:
        for wave, H, W, J in (('db4', 70, 66, 3), ('db10', 64, 64, 3), ('db2', 36, 52, 2), ('db6', 96, 160, 3), ('db8', 128, 128, 2)):
            x = torch.tensor(rng.randn(2, 2, H, W), dtype=torch.float32, device=dev)
            xfm = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev)
            ifm = pw.DWTInverse(wave=wave, mode='periodization').to(dev)
            yl, yh = xfm(x)
            y = ifm((yl, yh))
            oy = wo.dwt_inverse(yl.cpu().double().numpy(), [t.cpu().double().numpy() for t in yh], _flat(ifm.g0_col), _flat(ifm.g1_col),
                                _flat(ifm.g0_row), _flat(ifm.g1_row), 'periodization')
            assert _rel(y, oy) <= 1e-5, (wave, H, W, J, _rel(y, oy))
            if min(H, W) >> (J - 1) >= ifm.g0_col.numel():   # (a level shorter than the filter is no perfect-reconstruction pair in the reference either)
                assert float((y[..., :H, :W] - x).abs().max()) <= 2e-5 * float(x.abs().max()), (wave, H, W, J)
            yh2 = list(yh)
            yh2[0] = None                      # a band the caller left out: zeros (dwt/transform2d.py:137-139)
            y2 = ifm((yl, yh2))
            oy2 = wo.dwt_inverse(yl.cpu().double().numpy(), [None if t is None else t.cpu().double().numpy() for t in yh2], _flat(ifm.g0_col),
                                 _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), 'periodization')
            assert _rel(y2, oy2) <= 1e-5, (wave, 'None band')
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML = prev
