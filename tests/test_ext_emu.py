"""CPU tests (host emulation of the kernels): 1-D DWT modules, the stationary transform, the DTCWT 1-D primitives and
the function-level banks against the goldens generated from the real reference (oracle/pin_extras.py)."""
import numpy as np
import pytest
import torch

import _ext_cases as E
import _golden as G
import emu_backend
from oracle import wavelet_oracle as wo


@pytest.mark.parametrize('name', E.DWT1D_CASES)
def test_dwt1d_modules_and_gradients(name):
    with emu_backend.emulated():
        E.check_dwt1d(name, 'cpu', torch.float64, 5e-7)
        E.check_dwt1d(name, 'cpu', torch.float32, 1e-5)


@pytest.mark.parametrize('name', E.SWT_CASES)
def test_swt_level_and_dilated_bank(name):
    with emu_backend.emulated():
        E.check_swt(name, 'cpu', torch.float64, 5e-7)


def test_swt_second_level_is_the_dilated_bank_on_ll_and_bad_modes_raise():
    """J > 1 (which upstream cannot run): level 2 = the a-trous bank with dilation 2 on the ll channels of level 1
    (checked against the oracle); the default mode 'periodization' raises like upstream's mypad."""
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt.transform2d import SWTForward
    rng = np.random.RandomState(3)
    x = rng.randn(1, 2, 20, 24)
    h0, h1 = filters.dwt_analysis_taps('db2')
    torch.set_default_dtype(torch.float64)
    try:
        with emu_backend.emulated():
            y = SWTForward(J=2, wave='db2', mode='periodic')(torch.tensor(x))
            with pytest.raises(ValueError, match='Unkown pad type'):
                SWTForward(J=1, wave='db2')(torch.tensor(x))
    finally:
        torch.set_default_dtype(torch.float32)
    o1 = wo.afb2d_atrous(x, h0, h1, h0, h1, 'periodic', 1)
    o2 = wo.afb2d_atrous(o1[:, 0::4], h0, h1, h0, h1, 'periodic', 2)
    assert np.abs(y[0].numpy() - o1).max() < 1e-12 and np.abs(y[1].numpy() - o2).max() < 1e-12


SWT_KERNEL_CASES = [('db2', 'periodic', 1, (1, 2, 20, 24), torch.float64), ('db2', 'symmetric', 2, (2, 1, 37, 70), torch.float32),
                    ('db4', 'reflect', 4, (1, 1, 45, 130), torch.float32), ('haar', 'zero', 1, (1, 3, 5, 7), torch.float32),
                    ('db3', 'replicate', 2, (1, 2, 33, 65), torch.float16), ('db7', 'constant', 1, (1, 1, 40, 64), torch.float32),
                    ('db10', 'symmetric', 2, (1, 1, 36, 30), torch.float32), ('db4', 'symmetric', 2, (1, 1, 36, 30), torch.float64), ('db5', 'periodic', 3, (1, 1, 16, 200), torch.float32)]


@pytest.mark.parametrize('wave,mode,dil,shape,dtype', SWT_KERNEL_CASES)
def test_swt_level_kernel_vs_oracle(wave, mode, dil, shape, dtype):
    """wl_swt2d_level (csrc/wl_swt2d.h: one launch per level) against the oracle: every pad mode of the reference's mypad,
    dilations 1-4 (3: an odd one), tiles with ragged edges, compile-time and run-time tap counts (db7 = 14 taps), the three
    storage types, and the ll channels of a previous level as a strided view (no copy)."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt import lowlevel as ll
    rng = np.random.RandomState(11)
    h0, h1 = filters.dwt_analysis_taps(wave)
    N, C, H, W = shape
    big = rng.randn(N, 4 * C, H, W)
    tol = {torch.float64: 1e-12, torch.float32: 2e-6, torch.float16: 3e-3}[dtype]
    with emu_backend.emulated():
        filts = tuple(torch.tensor(np.asarray(v)) for v in (h0, h1, h0, h1))          # the stored (reversed) taps, as tensors
        xb = torch.tensor(big).to(dtype)
        for x in (xb[:, :C].contiguous(), xb[:, 0::4]):                  # dense planes; every 4th plane of a level's output
            c0 = pw.launch_count()
            y = ll.afb2d_atrous(x, filts, mode, dil)
            assert pw.kernels_since(c0)[0].startswith('WlSwtLevel'), pw.kernels_since(c0)
            ref = wo.afb2d_atrous(x.double().numpy(), h0, h1, h0, h1, 'zero' if mode == 'constant' else mode, dil)
            assert y.shape == ref.shape and y.dtype == dtype
            assert np.abs(y.double().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())


def test_swt_level_kernel_declines_what_it_does_not_cover():
    """Dilated filters too long for a tile in LDS run on the single-axis kernels, with the same result."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt import lowlevel as ll
    h0, h1 = filters.dwt_analysis_taps('db10')
    x = np.random.RandomState(2).randn(1, 1, 24, 40)
    with emu_backend.emulated():
        c0 = pw.launch_count()
        y = ll.afb2d_atrous(torch.tensor(x), tuple(torch.tensor(np.asarray(v)) for v in (h0, h1, h0, h1)), 'periodic', 8)
        assert not any(k.startswith('WlSwtLevel') for k in pw.kernels_since(c0))
    assert np.abs(y.numpy() - wo.afb2d_atrous(x, h0, h1, h0, h1, 'periodic', 8)).max() < 1e-11


@pytest.mark.parametrize('mode,shape,dtype', [('symmetric', (2, 3, 70, 134), torch.float32), ('zero', (1, 2, 36, 66), torch.float64),
                                              ('symmetric', (1, 1, 20, 18), torch.float16), ('symmetric', (1, 2, 34, 64), torch.float64)])
def test_rot_level1_kernel_vs_oracle(mode, shape, dtype):
    """wl_dtcwt_fwd_level1_rot (csrc/wl_dtcwt_rot.h: the seven single-axis filters + q2c of fwd_j1_rot in one launch) against
    the oracle's single-axis filters: both pad modes, ragged tiles, the (N, 6, C, h, w) layout for several planes, and the
    ScatLayer epilogue (scat = 1) against the magnitudes computed from the oracle's coefficients."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters, ops
    from pytorch_wavelets_amd.dtcwt import lowlevel as dl
    from pytorch_wavelets_amd.dtcwt import transform_funcs as tf
    h0o, _, h1o, _, h2o, _ = filters.biort('near_sym_b_bp')
    x = np.random.RandomState(8).randn(*shape)
    tol = {torch.float64: 1e-12, torch.float32: 3e-6, torch.float16: 4e-3}[dtype]
    h = [dl.prep_filt(v, 1).to(torch.float64) for v in (h0o, h1o, h2o)]
    hn = [v.numpy().ravel() for v in h]
    ll, re, im = E.rot_level1_reference(x if dtype != torch.float16 else x.astype(np.float16).astype(np.float64), *hn, mode)
    with emu_backend.emulated():
        xt = torch.tensor(x).to(dtype)
        c0 = pw.launch_count()
        l2, r2, i2 = tf.fwd_j1_rot(xt, *h, False, 1, mode)
        assert pw.kernels_since(c0) == ['WlDtFwd1Rot<%s, 0>' % {torch.float64: 'double', torch.float32: 'float', torch.float16: '_Float16'}[dtype]]
        for got, want in ((l2, ll), (r2, re), (i2, im)):
            assert got.shape == want.shape and got.dtype == dtype
            assert np.abs(got.double().numpy() - want).max() <= tol * max(1.0, np.abs(want).max())
        z = ops.dtcwt_fwd1_rot(xt, *h, mode == 'symmetric', scat=True, magbias=0.01)
        pool = ll.reshape(ll.shape[0], ll.shape[1], ll.shape[2] // 2, 2, ll.shape[3] // 2, 2).mean(axis=(3, 5))
        want = np.concatenate([pool[:, None], np.sqrt(re ** 2 + im ** 2 + 1e-4) - 0.01], 1)
        assert z.shape == want.shape and np.abs(z.double().numpy() - want).max() <= tol * max(1.0, np.abs(want).max())
        # odd sizes: declined (the modules pad first), the single-axis path answers
        assert ops.dtcwt_fwd1_rot(torch.randn(1, 1, 9, 8).to(dtype), *h, True) is None


@pytest.mark.parametrize('name', E.NONSEP_CASES)
def test_nonseparable_banks(name):
    with emu_backend.emulated():
        E.check_nonsep(name, 'cpu', torch.float64, 5e-7)
        E.check_nonsep(name, 'cpu', torch.float32, 1e-5)


def test_oracle_nonseparable_vs_reference_goldens():
    for name in E.NONSEP_CASES:
        meta, g = G.INDEX[name], G.load(name)
        if meta['mode'] != 'periodic':
            assert G.relerr(wo.afb2d_nonsep(g['x'].astype(np.float64), g['fa'].astype(np.float64), meta['mode']), g, 'y') < 5e-6
        assert G.relerr(wo.sfb2d_nonsep(g['c'].astype(np.float64), g['fs'].astype(np.float64), meta['mode']), g, 'rec') < 5e-6


def test_dtcwt_primitives():
    with emu_backend.emulated():
        E.check_prims('cpu', torch.float64, 5e-7)
        E.check_prims('cpu', torch.float32, 1e-5)


def test_function_level_afb1d_sfb1d():
    with emu_backend.emulated():
        E.check_afb1d_functions('cpu', 1e-5)


def test_function_level_periodization_odd_taps_and_short_signals():
    with emu_backend.emulated():
        E.check_afb1d_periodization('cpu', torch.float64, 1e-12)
        E.check_afb1d_periodization('cpu', torch.float32, 2e-5)


def test_oracle_extras_vs_reference_goldens():
    """The numpy oracle's restatement of the 1-D DWT / a-trous bank / primitives reproduces the reference goldens."""
    from pytorch_wavelets_amd import filters
    for name in E.DWT1D_CASES:
        meta, g = G.INDEX[name], G.load(name)
        h0, h1 = filters.dwt_analysis_taps(meta['wave'])
        g0, g1 = filters.dwt_synthesis_taps(meta['wave'])
        yl, yh = wo.dwt1d_forward(g['x'].astype(np.float64), meta['J'], h0, h1, meta['mode'])
        assert G.relerr(yl, g, 'yl') < 5e-7
        assert G.relerr(wo.dwt1d_inverse(yl, yh, g0, g1, meta['mode']), g, 'rec') < 5e-7
    for name in E.SWT_CASES:
        meta, g = G.INDEX[name], G.load(name)
        h0, h1 = filters.dwt_analysis_taps(meta['wave'])
        assert G.relerr(wo.afb2d_atrous(g['x'].astype(np.float64), h0, h1, h0, h1, meta['mode'], 1), g, 'y') < 5e-7


@pytest.mark.parametrize('name', E.SCATJ2_CASES)
def test_scatlayerj2_forward_and_backward(name):
    """ScatLayerj2 (two fused ScatLayer launches + the level-2 DTCWT kernel) against the reference's forward and its
    hand-written backward, incl. combine_colour and sizes that are not multiples of 8."""
    with emu_backend.emulated():
        E.check_scatj2(name, 'cpu', torch.float64, 5e-7)
        E.check_scatj2(name, 'cpu', torch.float32, 2e-5)


def test_scatlayerj2_refuses_what_upstream_refuses():
    import pytorch_wavelets_amd as pw
    with emu_backend.emulated():
        with pytest.raises(NotImplementedError):
            pw.ScatLayerj2(mode='zero')(torch.randn(1, 1, 16, 16))   # upstream: rowdfilt knows only 'symmetric'
    with pytest.raises(AssertionError):
        pw.ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_a')     # upstream asserts the matching q-shift family


def test_oracle_scatlayerj2_vs_reference_goldens():
    from pytorch_wavelets_amd import filters
    for name in E.SCATJ2_CASES:
        meta, g = G.INDEX[name], G.load(name)
        h0o, _, h1o, _ = filters.biort('near_sym_a')[:4]
        h0a, h0b, _, _, h1a, h1b, _, _ = filters.qshift('qshift_a')[:8]
        f = [np.asarray(v, dtype=np.float64).ravel()[::-1] for v in (h0o, h1o, h0a, h0b, h1a, h1b)]
        Z = wo.scat_layer_j2_forward(g['x'].astype(np.float64), *f, combine_colour=meta['combine_colour'])
        assert G.relerr(Z, g, 'Z') < 5e-7


@pytest.mark.parametrize('name', E.ROT_CASES)
def test_rotationally_symmetric_variants(name):
    """ScatLayer / ScatLayerj2 with biort='near_sym_b_bp' (third band-pass pair for the diagonal orientations):
    forward and backward against the reference."""
    with emu_backend.emulated():
        E.check_rot(name, 'cpu', torch.float64, 5e-7)
        E.check_rot(name, 'cpu', torch.float32, 2e-5)


def test_fused_multilevel_dwt1d_vs_oracle():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    try:
        with emu_backend.emulated():
            E.check_dwt1d_fused('cpu', tol=3e-6)
    finally:
        torch.set_default_dtype(prev)


def test_dwt1d_deep_pyramids_in_groups_of_four():
    """J = 6: DWT1DForward runs its levels four + two, DWT1DInverse two + four from the coarse end (SFB1DMulti per group, a `None`
    level inside a group on the per-level path); forward, inverse and the gradient of the inverse against the oracle / the
    per-level path."""
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import filters
    from pytorch_wavelets_amd.dwt import lowlevel as ll
    rng = np.random.RandomState(4)
    x = rng.randn(2, 3, 3000)
    h0, h1 = filters.dwt_analysis_taps('db3')
    g0, g1 = filters.dwt_synthesis_taps('db3')
    oyl, oyh = wo.dwt1d_forward(x, 6, h0, h1, 'symmetric')
    orec = wo.dwt1d_inverse(oyl, oyh, g0, g1, 'symmetric')
    with emu_backend.emulated():
        xfm, ifm = pw.DWT1DForward(J=6, wave='db3', mode='symmetric').float(), pw.DWT1DInverse(wave='db3', mode='symmetric').float()
        yl, yh = xfm(torch.tensor(x, dtype=torch.float32))
        assert np.abs(yl.numpy() - oyl).max() < 2e-5 * np.abs(oyl).max()
        ylr = yl.clone().requires_grad_(True)
        yhr = [h.clone().requires_grad_(True) for h in yh]
        rec = ifm((ylr, yhr))
        assert rec.shape == orec.shape and np.abs(rec.detach().numpy() - orec).max() < 5e-5 * np.abs(orec).max()
        gr = torch.autograd.grad((rec * rec).sum(), [ylr] + yhr)
        ll.FUSED_LEVELS = False
        try:
            yl2 = yl.clone().requires_grad_(True)
            yh2 = [h.clone().requires_grad_(True) for h in yh]
            rec2 = ifm((yl2, yh2))
            gr2 = torch.autograd.grad((rec2 * rec2).sum(), [yl2] + yh2)
        finally:
            ll.FUSED_LEVELS = True
        assert float((rec - rec2).abs().max()) < 1e-5 * float(rec2.abs().max())
        for a, b in zip(gr, gr2):
            assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))
        # (a `None` level is zeros of the LOWPASS length, upstream too: only where that equals the level's own length - here the
        # coarsest - does the pyramid still fit together)
        rec_none = ifm((yl, list(yh[:5]) + [None]))
        want = wo.dwt1d_inverse(oyl, list(oyh[:5]) + [None], g0, g1, 'symmetric')
        assert rec_none.shape == want.shape and np.abs(rec_none.numpy() - want).max() < 5e-5 * max(1.0, np.abs(want).max())
