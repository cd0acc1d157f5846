"""-m gpu: the single-axis kernels (wl_corr1d / wl_synth1d) behind DWT1DForward / DWT1DInverse, SWTForward, the
function-level banks and the DTCWT 1-D primitives, through the C ABI on the MI355X, against the reference goldens."""
import numpy as np
import pytest
import torch

import _ext_cases as E
from oracle import wavelet_oracle as wo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('name', E.DWT1D_CASES)
def test_dwt1d_modules_and_gradients(name):
    E.check_dwt1d(name, DEV, torch.float32, 1e-5)
    E.check_dwt1d(name, DEV, torch.float64, 5e-7)


@pytest.mark.parametrize('name', E.SWT_CASES)
def test_swt_level_and_dilated_bank(name):
    E.check_swt(name, DEV, torch.float32, 1e-5)


@pytest.mark.parametrize('wave,mode,dil,shape,dtype', [('db2', 'periodic', 2, (3, 2, 131, 200), torch.float32),
                                                       ('db4', 'symmetric', 4, (2, 3, 256, 256), torch.float32),
                                                       ('db3', 'replicate', 1, (2, 2, 70, 330), torch.float16),
                                                       ('db7', 'reflect', 2, (1, 2, 96, 129), torch.float64)])
def test_swt_level_kernel_vs_oracle_and_single_axis_path(wave, mode, dil, shape, dtype):
    """wl_swt2d_level (one launch per level, csrc/wl_swt2d.h) through the C ABI: against the oracle and against the
    single-axis path it replaces, on dense planes and on the ll channels of a previous level (a strided view)."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt import lowlevel as dwl
    h0, h1 = filters.dwt_analysis_taps(wave)
    N, C, H, W = shape
    torch.manual_seed(4)
    xb = torch.randn(N, 4 * C, H, W, device=DEV).to(dtype)
    filts = tuple(torch.tensor(np.asarray(v), device=DEV) for v in (h0, h1, h0, h1))
    tol = {torch.float64: 1e-11, torch.float32: 1e-5, torch.float16: 3e-3}[dtype]
    for x in (xb[:, :C].contiguous(), xb[:, 0::4]):
        c0 = pw.launch_count()
        y = dwl.afb2d_atrous(x, filts, mode, dil)
        assert pw.kernels_since(c0)[0].startswith('WlSwtLevel'), pw.kernels_since(c0)
        ref = wo.afb2d_atrous(x.double().cpu().numpy(), h0, h1, h0, h1, mode, dil)
        assert np.abs(y.double().cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())
        dwl.FUSED_LEVELS = False
        try:
            y2 = dwl.afb2d_atrous(x, filts, mode, dil)
        finally:
            dwl.FUSED_LEVELS = True
        assert float((y.double() - y2.double()).abs().max()) <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('mode,shape,dtype', [('symmetric', (3, 2, 130, 200), torch.float32), ('zero', (2, 3, 64, 64), torch.float32),
                                              ('symmetric', (2, 2, 96, 258), torch.float16), ('symmetric', (1, 2, 50, 66), torch.float64)])
def test_rot_level1_kernel_vs_oracle_and_single_axis_path(mode, shape, dtype):
    """wl_dtcwt_fwd_level1_rot (one launch, csrc/wl_dtcwt_rot.h) through the C ABI: against the oracle's single-axis filters and
    against the seven-launch path it replaces; the ScatLayer epilogue (scat = 1) against the module's differentiable chain."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dtcwt import lowlevel as dl
    from pytorch_wavelets_amd.dtcwt import transform_funcs as tf
    h0o, _, h1o, _, h2o, _ = filters.biort('near_sym_b_bp')
    torch.manual_seed(6)
    x = torch.randn(*shape, device=DEV).to(dtype)
    h = [dl.prep_filt(v, 1).to(DEV).to(torch.float64) for v in (h0o, h1o, h2o)]
    tol = {torch.float64: 1e-11, torch.float32: 1e-5, torch.float16: 4e-3}[dtype]
    want = E.rot_level1_reference(x.double().cpu().numpy(), *[v.cpu().numpy().ravel() for v in h], mode)
    c0 = pw.launch_count()
    got = tf.fwd_j1_rot(x, *h, False, 1, mode)
    assert pw.kernels_since(c0)[0].startswith('WlDtFwd1Rot'), pw.kernels_since(c0)
    tf.FUSED_ROT = False
    try:
        old = tf.fwd_j1_rot(x, *h, False, 1, mode)
    finally:
        tf.FUSED_ROT = True
    for g, w, o in zip(got, want, old):
        scale = max(1.0, np.abs(w).max())
        assert np.abs(g.double().cpu().numpy() - w).max() <= tol * scale
        assert float((g.double() - o.double()).abs().max()) <= 2 * tol * scale
    if dtype != torch.float64:
        m = pw.ScatLayer(biort='near_sym_b_bp', mode=mode).to(DEV).to(dtype)
        with torch.no_grad():
            c0 = pw.launch_count()
            z = m(x)
            assert pw.kernels_since(c0) == ['WlDtFwd1Rot<%s, 1>' % ('float' if dtype == torch.float32 else '_Float16')]
        z2 = m(x.clone().requires_grad_(True))                       # the differentiable chain (kernel + tensor-library magnitudes)
        assert float((z.float() - z2.float()).abs().max()) <= 2 * tol * max(1.0, float(z2.float().abs().max()))


@pytest.mark.parametrize('name', E.NONSEP_CASES)
def test_nonseparable_banks(name):
    E.check_nonsep(name, DEV, torch.float32, 1e-5)
    E.check_nonsep(name, DEV, torch.float64, 5e-7)


def test_nonseparable_equals_separable_at_size():
    """afb2d_nonsep with the outer-product point-spread functions of db4 == the separable analysis kernel, and
    sfb2d_nonsep inverts it, on a 2x3x256x320 batch (symmetric)."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt import lowlevel as dwl
    w = filters.Wavelet('db4')
    x = torch.randn(2, 3, 256, 320, device=DEV)
    y = dwl.afb2d_nonsep(x, (w.dec_lo, w.dec_hi), 'symmetric')
    yl, yh = pw.DWTForward(J=1, wave='db4', mode='symmetric').to(DEV)(x)
    y5 = y.reshape(2, 3, 4, y.shape[-2], y.shape[-1])
    assert float((y5[:, :, 0] - yl).abs().max()) < 1e-5 * float(yl.abs().max())
    assert float((y5[:, :, 1:] - yh[0]).abs().max()) < 1e-5 * float(yh[0].abs().max())
    rec = dwl.sfb2d_nonsep(y5, (w.rec_lo, w.rec_hi), 'symmetric')
    assert float((rec[..., :256, :320] - x).abs().max()) < 1e-4


def test_dtcwt_primitives():
    E.check_prims(DEV, torch.float32, 1e-5)
    E.check_prims(DEV, torch.float64, 5e-7)


def test_function_level_afb1d_sfb1d():
    E.check_afb1d_functions(DEV, 1e-5)


def test_function_level_periodization_odd_taps_and_short_signals():
    E.check_afb1d_periodization(DEV, torch.float64, 1e-12)
    E.check_afb1d_periodization(DEV, torch.float32, 2e-5)


def test_dwt1d_long_signals_vs_oracle_and_fp16():
    """Long rows (several workgroups per signal), odd length, float16 storage."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    torch.manual_seed(11)
    x = torch.randn(3, 5, 100001)
    h0, h1 = filters.dwt_analysis_taps('db6')
    g0, g1 = filters.dwt_synthesis_taps('db6')
    oyl, oyh = wo.dwt1d_forward(x.double().numpy(), 3, h0, h1, 'symmetric')
    xfm, ifm = pw.DWT1DForward(J=3, wave='db6', mode='symmetric').to(DEV), pw.DWT1DInverse(wave='db6', mode='symmetric').to(DEV)
    yl, yh = xfm(x.to(DEV))
    rel = lambda a, b: float(np.abs(a.double().cpu().numpy() - b).max() / np.abs(b).max())   # noqa: E731
    assert rel(yl, oyl) < 1e-5 and all(rel(a, b) < 1e-5 for a, b in zip(yh, oyh))
    rec = ifm((yl, yh))
    assert float((rec[..., :100001].cpu() - x).abs().max()) < 1e-4
    yl16, yh16 = xfm.half()(x.half().to(DEV))
    assert yl16.dtype == torch.float16 and rel(yl16, oyl) < 5e-3


@pytest.mark.parametrize('name', E.SCATJ2_CASES)
def test_scatlayerj2_forward_and_backward(name):
    E.check_scatj2(name, DEV, torch.float32, 2e-5)
    E.check_scatj2(name, DEV, torch.float64, 5e-7)


@pytest.mark.parametrize('name', E.ROT_CASES)
def test_rotationally_symmetric_variants(name):
    E.check_rot(name, DEV, torch.float32, 2e-5)
    E.check_rot(name, DEV, torch.float64, 5e-7)


def test_fused_multilevel_dwt1d_vs_oracle_gpu():
    E.check_dwt1d_fused(DEV)
    E.check_dwt1d_fused(DEV, cases=[('db4', 'symmetric', 3, (8, 16, 65536), torch.float32), ('db8', 'zero', 4, (4, 3, 100003), torch.float32),
                                    ('db4', 'symmetric', 3, (8, 16, 65536), torch.float16)])
