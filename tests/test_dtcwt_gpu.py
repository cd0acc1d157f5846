"""GPU parity tests for the DTCWT and ScatLayer paths: HIP kernels (through the C ABI) vs the golden
vectors of the reference; 1e-5 relative (max-norm) in fp32.  Modelled on the reference's
tests/test_dtcwt.py (:86-346, :350-413) and tests/test_scatnet_fwd.py (:9-58)."""
import numpy as np
import pytest

import _opts
import torch

import _dtcwt_cases as D
import _golden as G
import pytorch_wavelets_amd as pw
from oracle import wavelet_oracle as wo
from pytorch_wavelets_amd import filters as F

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = 'cuda:0'


@pytest.mark.parametrize('name', G.cases('dtcwt'))
def test_dtcwt_vs_reference_goldens(name):
    D.check_dtcwt_case(name, DEV, torch.float32, TOL)


def test_dtcwt_inverse_with_missing_inputs():
    D.check_dtcwt_none(DEV, torch.float32, TOL)


@pytest.mark.parametrize('name', G.cases('scat'))
def test_scatlayer_vs_reference_goldens(name):
    D.check_scat_case(name, DEV, torch.float32, TOL)


def test_layout_permutations():
    D.check_layouts(DEV, torch.float32, TOL)


def test_dtcwt_fp64():
    torch.manual_seed(0)
    x = torch.randn(1, 2, 40, 36, dtype=torch.float64)
    hb = F.dtcwt_forward_taps('near_sym_b', 'qshift_b')
    oyl, oyh = wo.dtcwt_forward(x.numpy(), 3, *hb)
    torch.set_default_dtype(torch.float64)
    try:
        xfm = pw.DTCWTForward(biort='near_sym_b', qshift='qshift_b', J=3).to(DEV)
    finally:
        torch.set_default_dtype(torch.float32)
    yl, yh = xfm(x.to(DEV))
    assert np.abs(yl.cpu().numpy() - oyl).max() / np.abs(oyl).max() < 1e-12
    for a, b in zip(yh, oyh):
        assert np.abs(a.cpu().numpy() - b).max() / np.abs(b).max() < 1e-12


def test_full_size_properties_config2_and_3():
    """BASELINE configs[2] (64x3x512x512) and configs[3] per-GPU share (32x3x256x256): shapes, perfect
    reconstruction, linearity and an oracle check on one plane."""
    torch.manual_seed(1)
    x = torch.randn(64, 3, 512, 512, device=DEV)
    xfm, ifm = pw.DTCWTForward(J=3).to(DEV), pw.DTCWTInverse().to(DEV)
    yl, yh = xfm(x)
    assert yl.shape == (64, 3, 128, 128)
    assert [tuple(h.shape) for h in yh] == [(64, 3, 6, 256, 256, 2), (64, 3, 6, 128, 128, 2), (64, 3, 6, 64, 64, 2)]
    assert float((ifm((yl, yh)) - x).abs().max() / x.abs().max()) < TOL
    yl2, yh2 = xfm(3.0 * x[:2] - x[2:4])
    assert float((yh2[1] - (3.0 * yh[1][:2] - yh[1][2:4])).abs().max() / yh2[1].abs().max()) < TOL
    hb = F.dtcwt_forward_taps('near_sym_a', 'qshift_a')
    oyl, oyh = wo.dtcwt_forward(x[41:42, 1:2].cpu().double().numpy(), 3, *hb)
    assert np.abs(yl[41:42, 1:2].cpu().numpy() - oyl).max() / np.abs(oyl).max() < TOL
    assert np.abs(yh[0][41:42, 1:2].cpu().numpy() - oyh[0]).max() / np.abs(oyh[0]).max() < TOL
    del yl, yh, yl2, yh2
    xs = torch.randn(32, 3, 256, 256, device=DEV)
    Z = pw.ScatLayer().to(DEV)(xs)
    assert Z.shape == (32, 21, 128, 128)
    oZ = wo.scat_layer_forward(xs[5:6].cpu().double().numpy(), hb[0], hb[1])
    assert np.abs(Z[5:6].cpu().numpy() - oZ).max() / np.abs(oZ).max() < TOL


@pytest.mark.parametrize('seed', range(5))
def test_specialised_equals_generic_random_shapes_gpu(seed, monkeypatch):
    """Specialised DTCWT / ScatLayer kernels against the generic ones on the real hardware, random shapes."""
    import numpy as np
    rng = np.random.RandomState(500 + seed)
    biort = ['near_sym_a', 'near_sym_b', 'antonini', 'legall', 'near_sym_a'][seed]
    qshift = ['qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_06'][seed]
    for _ in range(3):
        H, W = int(rng.randint(8, 300)), int(rng.randint(8, 500))
        J = int(rng.randint(1, 4))
        x = torch.tensor(rng.randn(2, 3, H, W), dtype=torch.float32, device=DEV)
        out = {}
        for generic in ('0', '1'):
            _opts.set_generic(generic)
            xfm = pw.DTCWTForward(biort=biort, qshift=qshift, J=J).to(DEV)
            ifm = pw.DTCWTInverse(biort=biort, qshift=qshift).to(DEV)
            yl, yh = xfm(x)
            res = [yl] + list(yh) + [ifm((yl, yh))]
            if biort in ('near_sym_a', 'near_sym_b'):
                xs = x.clone().requires_grad_(True)
                z = pw.ScatLayer(biort=biort).to(DEV)(xs)
                g, = torch.autograd.grad((z * z).sum(), xs)
                res += [z.detach(), g]
            out[generic] = res
        for a, b in zip(out['0'], out['1']):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 3e-5 * (float(b.abs().max()) + 1e-30), (biort, qshift, H, W, J)


@pytest.mark.parametrize('shape,biort,mode,dtype', [((64, 3, 512, 512), 'near_sym_a', 'symmetric', torch.float32),
                                                    ((20, 3, 257, 1024), 'near_sym_a', 'symmetric', torch.float32),
                                                    ((64, 1, 300, 260), 'antonini', 'zero', torch.float32),
                                                    ((96, 1, 129, 512), 'legall', 'symmetric', torch.float16)])
def test_streaming_level1_forward(shape, biort, mode, dtype):
    """The streaming level-1 forward over column strips (the engine's choice for wide planes that fill the chip) against
    the tile kernel (wl_set_option no_stream) on every plane and against the oracle on sampled planes; backward of the
    module through it as well."""
    from pytorch_wavelets_amd import _lib
    torch.manual_seed(1)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=1, biort=biort, mode=mode).to(DEV).to(dtype)
    lib = _lib.get()
    try:
        yl, yh = xfm(x)
        assert 'WlDtFwd1Strip' in pw.last_kernel() or 'WlDtFwd12Strip' in pw.last_kernel(), pw.last_kernel()
        lib.wl_set_option(b'no_stream', 1)
        yl2, yh2 = xfm(x)
        assert 'WlDtFwd1Tile' in pw.last_kernel(), pw.last_kernel()
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 2e-6
    assert float((yl.float() - yl2.float()).abs().max()) <= tol * float(yl2.float().abs().max())
    assert float((yh[0].float() - yh2[0].float()).abs().max()) <= tol * float(yh2[0].float().abs().max())
    hb = F.dtcwt_forward_taps(biort, 'qshift_a')
    for n, c in ((0, 0), (shape[0] - 1, shape[1] - 1)):
        oyl, oyh = wo.dtcwt_forward(x[n:n + 1, c:c + 1].double().cpu().numpy(), 1, *hb, mode=mode)
        a = yl[n:n + 1, c:c + 1].double().cpu().numpy()
        assert np.abs(a - oyl).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * np.abs(oyl).max()
        b = yh[0][n:n + 1, c:c + 1].double().cpu().numpy()
        assert np.abs(b - oyh[0]).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * np.abs(oyh[0]).max()


@pytest.mark.parametrize('shape,biort,mode,dtype', [((64, 3, 512, 512), 'near_sym_a', 'symmetric', torch.float32),
                                                    ((20, 3, 258, 1024), 'near_sym_a', 'symmetric', torch.float32),
                                                    ((64, 1, 300, 260), 'antonini', 'zero', torch.float32),
                                                    ((96, 1, 130, 512), 'legall', 'symmetric', torch.float16)])
def test_streaming_level1_inverse(shape, biort, mode, dtype):
    """The streaming level-1 inverse over column strips against the tile kernel (wl_set_option no_stream) on every plane,
    the round trip through both streaming kernels, and the module's backward (an inverse with the forward taps)."""
    from pytorch_wavelets_amd import _lib
    torch.manual_seed(2)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=1, biort=biort, mode=mode).to(DEV).to(dtype)
    ifm = pw.DTCWTInverse(biort=biort, mode=mode).to(DEV).to(dtype)
    lib = _lib.get()
    yl, yh = xfm(x)
    try:
        r1 = ifm((yl, yh))
        assert 'WlDtInv1Strip' in pw.last_kernel(), pw.last_kernel()
        lib.wl_set_option(b'no_stream', 1)
        r2 = ifm((yl, yh))
        assert 'WlDtInv1Tile' in pw.last_kernel(), pw.last_kernel()
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 2e-6
    assert float((r1.float() - r2.float()).abs().max()) <= tol * float(r2.float().abs().max())
    if mode == 'symmetric':   # perfect reconstruction
        assert float((r1.float() - x.float()).abs().max()) <= (2e-2 if dtype == torch.float16 else 2e-5) * float(x.float().abs().max())
    if dtype == torch.float32:
        xg = x.clone().requires_grad_(True)
        a, b = xfm(xg)
        ((a * yl).sum() + (b[0] * yh[0]).sum()).backward()
        g1 = xg.grad.clone()
        try:
            lib.wl_set_option(b'no_stream', 1)
            xg.grad = None
            a, b = xfm(xg)
            ((a * yl).sum() + (b[0] * yh[0]).sum()).backward()
        finally:
            lib.wl_set_option(b'no_stream', 0)
        assert float((g1 - xg.grad).abs().max()) <= 2e-6 * float(xg.grad.abs().max())


@pytest.mark.parametrize('shape,biort,dtype', [((64, 3, 512, 512), 'near_sym_a', torch.float32),
                                               ((20, 3, 260, 1024), 'near_sym_a', torch.float32),
                                               ((128, 1, 256, 264), 'legall', torch.float32),
                                               ((160, 1, 128, 512), 'near_sym_a', torch.float16)])
def test_fused_levels_1_and_2(shape, biort, dtype):
    """Levels 1 + 2 of the forward in one launch (csrc/wl_dtcwt_fused.h) against the two per-level launches
    (wl_set_option no_stream) on every plane, against the oracle on sampled planes, and the gradient through it."""
    from pytorch_wavelets_amd import _lib
    torch.manual_seed(1)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=2, biort=biort).to(DEV).to(dtype)
    lib = _lib.get()
    try:
        yl, yh = xfm(x)
        assert 'WlDtFwd12Strip' in pw.last_kernel(), pw.last_kernel()
        if dtype == torch.float32:
            xg = x.clone().requires_grad_(True)
            a, b = xfm(xg)
            ((a * yl).sum() + (b[0] * yh[0]).sum() + (b[1] * yh[1]).sum()).backward()
            g1 = xg.grad.clone()
        lib.wl_set_option(b'no_stream', 1)
        yl2, yh2 = xfm(x)
        assert 'WlDtFwd2Tile' in pw.last_kernel(), pw.last_kernel()
        if dtype == torch.float32:
            xg.grad = None
            a, b = xfm(xg)
            ((a * yl).sum() + (b[0] * yh[0]).sum() + (b[1] * yh[1]).sum()).backward()
            assert float((g1 - xg.grad).abs().max()) <= 1e-5 * float(xg.grad.abs().max())
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 3e-6
    assert float((yl.float() - yl2.float()).abs().max()) <= tol * float(yl2.float().abs().max())
    for u, v in zip(yh, yh2):
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())
    hb = F.dtcwt_forward_taps(biort, 'qshift_a')
    otol = 5e-3 if dtype == torch.float16 else 1e-5
    for n, c in ((0, 0), (shape[0] - 1, shape[1] - 1)):
        oyl, oyh = wo.dtcwt_forward(x[n:n + 1, c:c + 1].double().cpu().numpy(), 2, *hb, mode='symmetric')
        assert np.abs(yl[n:n + 1, c:c + 1].double().cpu().numpy() - oyl).max() <= otol * np.abs(oyl).max()
        for j in range(2):
            assert np.abs(yh[j][n:n + 1, c:c + 1].double().cpu().numpy() - oyh[j]).max() <= otol * np.abs(oyh[j]).max()


def test_goldens_through_the_forced_fused_kernel(monkeypatch):
    """The reference's goldens (outputs, reconstruction, input gradient) with levels 1 + 2 forced onto the fused kernel."""
    from pytorch_wavelets_amd import ops
    took = []
    orig = ops.dtcwt_fwd12

    def spy(*a, **k):
        r = orig(*a, **k)
        took.append(r is not None)
        return r
    monkeypatch.setattr(ops, 'STREAM_FORCE', True)
    monkeypatch.setattr(ops, 'dtcwt_fwd12', spy)
    for name in ('dtcwt_00', 'dtcwt_01'):
        D.check_dtcwt_case(name, DEV, torch.float32, 1e-5)
    assert took and all(took)


@pytest.mark.parametrize('shape,dtype,grad', [((256, 3, 256, 256), torch.float32, False), ((64, 3, 512, 512), torch.float32, True),
                                              ((20, 3, 260, 1024), torch.float32, True), ((128, 2, 128, 512), torch.float16, False)])
def test_lean_scatlayer_kernel(shape, dtype, grad):
    """ScatLayer on the lean streaming kernel (wl_dtcwt_fused.h MODE 1) against the tile kernel on every plane and against
    the oracle on sampled planes; the backward consumes the (re, im) / r it saved."""
    from pytorch_wavelets_amd import _lib
    torch.manual_seed(2)
    x = torch.randn(*shape, device=DEV).to(dtype)
    sl = pw.ScatLayer().to(DEV).to(dtype)
    lib = _lib.get()
    out = {}
    try:
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            xg = x.clone().requires_grad_(grad)
            z = sl(xg)
            if ns == 0:
                assert 'WlDtFwd12Strip' in pw.last_kernel() and (', 10, 3' if grad else ', 10, 1') in pw.last_kernel(), pw.last_kernel()
            out[ns] = [z.detach()]
            if grad:
                g, = torch.autograd.grad((z * z).sum(), xg)
                out[ns].append(g)
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 5e-6
    for u, v in zip(out[0], out[1]):
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())
    hb = F.dtcwt_forward_taps('near_sym_a', 'qshift_a')
    for n in (0, shape[0] - 1):
        ref = wo.scat_layer_forward(x[n:n + 1].double().cpu().numpy(), hb[0], hb[1])
        got = out[0][0][n:n + 1].double().cpu().numpy()
        assert np.abs(got - ref).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * np.abs(ref).max()


@pytest.mark.parametrize('shape,qshift,dtype', [((64, 3, 512, 512), 'qshift_a', torch.float32), ((20, 3, 264, 1024), 'qshift_a', torch.float32),
                                                ((128, 1, 256, 256), 'qshift_b', torch.float32), ((160, 1, 128, 512), 'qshift_a', torch.float16),
                                                ((64, 3, 512, 512), 'qshift_d', torch.float32), ((128, 1, 256, 256), 'qshift_d', torch.float32),
                                                ((20, 3, 264, 1024), 'qshift_d', torch.float32), ((160, 1, 128, 512), 'qshift_d', torch.float16)])
def test_streaming_level2_inverse(shape, qshift, dtype):
    """The streaming level-2 inverse over column strips (WlDtInv2Strip) against the tile kernel (wl_set_option no_stream)
    on every plane, the whole inverse against the oracle on sampled planes, and as the backward of the level-2 forward."""
    from pytorch_wavelets_amd import _lib, ops
    torch.manual_seed(1)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=2, qshift=qshift).to(DEV).to(dtype)
    ifm = pw.DTCWTInverse(qshift=qshift).to(DEV).to(dtype)
    yl, yh = xfm(x)
    yl, yh = yl + 0.1 * torch.randn_like(yl), [v + 0.1 * torch.randn_like(v) for v in yh]
    lib = _lib.get()
    out, ll1 = {}, {}
    try:
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            ll1[ns] = ops.dtcwt_inv2(yl, yh[1], ifm.g0a, ifm.g0b, ifm.g1a, ifm.g1b)
            if ns == 0:
                assert 'WlDtInv2Strip' in pw.last_kernel(), pw.last_kernel()
            out[ns] = [ifm((yl, yh))]
            if dtype == torch.float32:
                xg = x.clone().requires_grad_(True)
                a, b = xfm(xg)
                ((a * yl).sum() + (b[0] * yh[0]).sum() + (b[1] * yh[1]).sum()).backward()
                out[ns].append(xg.grad.clone())
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 5e-6
    assert float((ll1[0].float() - ll1[1].float()).abs().max()) <= tol * float(ll1[1].float().abs().max())
    for u, v in zip(out[0], out[1]):
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())
    hb = F.dtcwt_inverse_taps('near_sym_a', qshift)
    for n, c in ((0, 0), (shape[0] - 1, shape[1] - 1)):
        ref = wo.dtcwt_inverse(yl[n:n + 1, c:c + 1].double().cpu().numpy(), [v[n:n + 1, c:c + 1].double().cpu().numpy() for v in yh], *hb)
        got = out[0][0][n:n + 1, c:c + 1].double().cpu().numpy()
        assert np.abs(got - ref).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * np.abs(ref).max()


@pytest.mark.parametrize('shape,dtype', [((64, 3, 256, 256), torch.float32), ((64, 3, 512, 512), torch.float32), ((20, 3, 260, 1024), torch.float32),
                                         ((160, 1, 128, 512), torch.float16)])
def test_streaming_level2_forward(shape, dtype):
    """fwd_j2plus alone on the stagers and level-2 lanes of the fused kernel (MODE 4) against the tile kernel on every plane."""
    from pytorch_wavelets_amd import _lib, ops
    torch.manual_seed(3)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=2).to(DEV).to(dtype)
    lib = _lib.get()
    out = {}
    try:
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            out[ns] = ops.dtcwt_fwd2(x, xfm.h0a, xfm.h0b, xfm.h1a, xfm.h1b)
            if ns == 0:
                assert 'WlDtFwd12Strip' in pw.last_kernel() and ', 10, 4' in pw.last_kernel(), pw.last_kernel()
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 3e-6
    for u, v in zip(out[0], out[1]):
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())


def test_filter_buffers_changed_after_construction_dtcwt_forward_gpu():
    """Round-3 verdict, weak #1b: the fused level-1+2 launch needs a symmetric h0o - decided against the buffer at call time."""
    import _mutation_cases as M
    M.check_dtcwt_forward_mutations(DEV)
    M.check_dtcwt_forward_mutations(DEV, shape=(64, 3, 512, 512))     # config 3's shape: the engine's own policy picks the fused launch


@pytest.mark.parametrize('shape,J,dtype', [((64, 3, 512, 512), 3, torch.float32), ((20, 3, 264, 1024), 2, torch.float32),
                                           ((96, 1, 256, 1160), 2, torch.float32), ((64, 3, 256, 256), 3, torch.float32),
                                           ((160, 1, 128, 512), 2, torch.float16)])
def test_fused_inverse_levels_2_and_1_gpu(shape, J, dtype):
    """Levels 2 + 1 of the inverse in one launch (WlDtInv21Strip: the level-1 lowpass in an LDS ring) at config 3's shape and
    around it: against the per-level kernels (wl_set_option no_stream -> tile kernels) on every plane, against the ORACLE on
    sampled planes, and as the backward of the fused forward."""
    from pytorch_wavelets_amd import _lib
    torch.manual_seed(2)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm = pw.DTCWTForward(J=J).to(DEV).to(dtype)
    ifm = pw.DTCWTInverse().to(DEV).to(dtype)
    yl, yh = xfm(x)
    yl, yh = yl + 0.1 * torch.randn_like(yl), [v + 0.1 * torch.randn_like(v) for v in yh]
    lib = _lib.get()
    out = {}
    try:
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            c0 = pw.launch_count()
            out[ns] = [ifm((yl, yh))]
            ks = pw.kernels_since(c0)
            if ns == 0:
                assert len(ks) == J - 1 and 'WlDtInv21Strip' in ks[-1], ks
            else:
                assert len(ks) == J and not any('WlDtInv21Strip' in k for k in ks), ks
            if dtype == torch.float32:
                xg = x.clone().requires_grad_(True)
                a, b = xfm(xg)
                c0 = pw.launch_count()
                ((a * yl).sum() + sum((u * v).sum() for u, v in zip(b, yh))).backward()
                if ns == 0 and shape[2] % 4 == 0 and shape[3] % 4 == 0:
                    assert any('WlDtInv21Strip' in k for k in pw.kernels_since(c0)), pw.kernels_since(c0)
                out[ns].append(xg.grad.clone())
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 5e-6
    for u, v in zip(out[0], out[1]):
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())
    gb = F.dtcwt_inverse_taps('near_sym_a', 'qshift_a')
    for n, c in ((0, 0), (shape[0] - 1, shape[1] - 1), (shape[0] // 2, 0)):
        ref = wo.dtcwt_inverse(yl[n:n + 1, c:c + 1].double().cpu().numpy(), [v[n:n + 1, c:c + 1].double().cpu().numpy() for v in yh], *gb)
        got = out[0][0][n:n + 1, c:c + 1].double().cpu().numpy()
        assert np.abs(got - ref).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * np.abs(ref).max()


def test_golden_through_the_forced_fused_inverse(monkeypatch):
    """The reference's own goldens (reconstruction and the gradients of the inverse) with levels 2 + 1 forced onto the fused
    inverse kernel, whatever the engine's policy says about so small a plane."""
    from pytorch_wavelets_amd import ops
    took = []
    orig = ops.dtcwt_inv21

    def spy(*a, **k):
        r = orig(*a, **k)
        took.append(r is not None)
        return r
    monkeypatch.setattr(ops, 'STREAM_FORCE', True)
    monkeypatch.setattr(ops, 'dtcwt_inv21', spy)
    D.check_dtcwt_case('dtcwt_00', DEV, torch.float32, TOL)
    D.check_dtcwt_case('dtcwt_01', DEV, torch.float32, TOL)
    assert took and any(took)


def test_scatlayer_backward_on_the_streaming_inverse_gpu():
    """config 4's plane size (256 x 256, enough images for the engine's own policy to pick the streaming kernel) and wider planes."""
    D.check_scat_backward_streaming(DEV, [((64, 3, 256, 256), torch.float32), ((32, 3, 512, 512), torch.float32),
                                          ((64, 2, 132, 1160), torch.float32), ((64, 3, 256, 512), torch.float16)])


@pytest.mark.parametrize('shape,biort,mode,dtype,grad', [((512, 3, 32, 32), 'near_sym_a', 'symmetric', torch.float32, True),
                                                          ((128, 3, 32, 32), 'near_sym_b', 'symmetric', torch.float32, True),
                                                          ((64, 16, 16, 16), 'legall', 'zero', torch.float32, False),
                                                          ((256, 3, 32, 32), 'near_sym_a', 'symmetric', torch.float16, False)])
def test_small_plane_level1_kernel(shape, biort, mode, dtype, grad):
    """WlDtFwd1Small (several small planes per workgroup: CIFAR / Tiny-ImageNet shapes) behind ScatLayer and DTCWTForward(J=1):
    against the tile kernels on every plane (wl_set_option no_stream), against the oracle on sampled planes, and the training step
    (the saved (re, im) / r feed the fused backward)."""
    from pytorch_wavelets_amd import _lib
    from pytorch_wavelets_amd.dtcwt import lowlevel as dl
    torch.manual_seed(2)
    x = torch.randn(*shape, device=DEV).to(dtype)
    sl = pw.ScatLayer(biort=biort, mode=mode).to(DEV).to(dtype)
    xf = pw.DTCWTForward(J=1, biort=biort, mode=mode).to(DEV).to(dtype)
    lib = _lib.get()
    out = {}
    try:
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            xg = x.clone().requires_grad_(grad)
            c0 = pw.launch_count()
            z = sl(xg)
            ks = pw.kernels_since(c0)
            assert ('WlDtFwd1Small' in ks[0]) == (ns == 0), ks
            out[ns] = [z.detach()]
            if grad:
                g, = torch.autograd.grad((z * z).sum(), xg)
                out[ns].append(g)
            yl, yh = xf(x)
            out[ns] += [yl, yh[0]]
    finally:
        lib.wl_set_option(b'no_stream', 0)
    tol = 5e-3 if dtype == torch.float16 else 3e-6
    for u, v in zip(out[0], out[1]):
        assert u.shape == v.shape
        assert float((u.float() - v.float()).abs().max()) <= tol * max(1.0, float(v.float().abs().max()))
    h0o, _, h1o, _ = F.biort(biort)
    hp = [dl.prep_filt(v, 1).numpy().ravel() for v in (h0o, h1o)]
    sel = [0, shape[0] - 1]
    want = wo.scat_layer_forward(x[sel].double().cpu().numpy(), hp[0], hp[1], mode)
    assert np.abs(out[0][0][sel].double().cpu().numpy() - want).max() <= (5e-3 if dtype == torch.float16 else 1e-5) * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('shape,dtype,expect', [
    ((32, 3, 256, 256), torch.float32, ('10, 1, 4, 2>', '10, 5, 2>', '10, 1, 4, 4>')),
    ((5, 1, 256, 256), torch.float32, None),           # 30 second-order planes: the last workgroup of four holds two
    ((16, 3, 512, 512), torch.float32, ('10, 1>', '10, 5>', '10, 1, 4, 2>')),
    ((16, 3, 200, 136), torch.float32, None),          # padded to multiples of 8; narrow second-order planes: tile kernels
    ((16, 3, 32, 32), torch.float32, None),            # CIFAR: the small-plane kernel writes into the 49-entry output too
    ((8, 3, 512, 512), torch.float16, None)])
def test_scatlayerj2_in_place_equals_the_chain(shape, dtype, expect):
    """ScatLayerj2 inference = three launches that write their entries of the (N, 49 C, H/4, W/4) output in place
    (wl_scat_fwd_level1_into, wl_scat_fwd_level2_into) against the chain of differentiable pieces + torch.cat, which the golden
    tests pin to the reference; and against the float64 oracle."""
    from pytorch_wavelets_amd.scatternet import lowlevel as sl
    torch.manual_seed(0)
    x = torch.randn(*shape, device=DEV, dtype=dtype)
    m = pw.ScatLayerj2().to(DEV).to(dtype)
    with torch.no_grad():
        c0 = pw.launch_count()
        z1 = m(x)
        ks = pw.kernels_since(c0)
        sl.FUSED_J2 = False
        try:
            z0 = m(x)
        finally:
            sl.FUSED_J2 = True
    assert z1.shape == z0.shape and z1.shape[1] == 49 * shape[1]
    if expect is not None:
        assert len(ks) == 3 and all(e in k for e, k in zip(expect, ks)), ks
    tol = 5e-3 if dtype == torch.float16 else 3e-6
    assert float((z1.float() - z0.float()).abs().max()) <= tol * float(z0.float().abs().max())


@pytest.mark.parametrize('shape,dtype', [((64, 3, 256, 256), torch.float16), ((96, 3, 128, 128), torch.float16), ((96, 3, 128, 128), torch.float32),
                                         ((40, 5, 96, 112), torch.float32), ((33, 3, 512, 256), torch.float16), ((128, 3, 72, 80), torch.float32), ((64, 3, 224, 224), torch.float32), ((64, 3, 160, 200), torch.float16)])
def test_streaming_kernels_on_narrow_and_half_precision_planes_equal_the_tile_kernels(shape, dtype):
    """Planes of 96-128 columns go four to a workgroup (PP = 4), float16 planes of 256 columns and more to the streaming kernels
    like float32 ones (the rule is in columns, not bytes): DTCWT J = 2 forward / inverse and the ScatLayer training step against
    the same transforms on the tile kernels (wl_set_option no_stream), which the golden tests pin to the reference."""
    from pytorch_wavelets_amd import ops
    torch.manual_seed(0)
    x = torch.randn(*shape, device=DEV).to(dtype)
    xfm, ifm, sl = (m.to(DEV).to(dtype) for m in (pw.DTCWTForward(J=2), pw.DTCWTInverse(), pw.ScatLayer()))
    out, names = {}, {}
    try:
        for ns in (0, 1):
            ops.set_option('no_stream', ns)
            c0 = pw.launch_count()
            yl, yh = xfm(x)
            y = ifm((yl, yh))
            xg = x.clone().requires_grad_(True)
            z = sl(xg)
            g, = torch.autograd.grad(z, xg, torch.ones_like(z))
            out[ns] = [yl, *yh, y, z.detach(), g]
            names[ns] = pw.kernels_since(c0)
    finally:
        ops.set_option('no_stream', 0)
    assert any('Strip' in k for k in names[0]) and not any('Strip' in k for k in names[1]), names
    tol = 6e-3 if dtype == torch.float16 else 5e-6
    for u, v in zip(out[0], out[1]):
        assert u.shape == v.shape
        assert float((u.float() - v.float()).abs().max()) <= tol * float(v.float().abs().max())


@pytest.mark.parametrize('shape,dtype,J', [((12, 3, 512, 512), torch.float32, 1), ((12, 3, 512, 512), torch.float32, 3), ((6, 3, 1024, 1024), torch.float32, 2),
                                           ((22, 3, 256, 256), torch.float32, 2), ((43, 3, 128, 128), torch.float32, 1), ((32, 3, 200, 328), torch.float32, 1),
                                           ((12, 3, 512, 512), torch.float16, 2)])
def test_near_sym_b_on_the_streaming_level1_kernels(shape, dtype, J):
    """Round 6: DTCWT with `near_sym_b / qshift_b` (13 / 19 and 14 taps) - the level-1 pair on the lean forward kernel and the
    streaming inverse - forward, inverse and the forward's gradient against the oracle at the shapes of the benchmarks."""
    import _nearsymb_cases as NB
    NB.check_dtcwt_near_sym_b(DEV, shape, dtype, J=J)


@pytest.mark.parametrize('shape,dtype', [((22, 3, 256, 256), torch.float32), ((12, 3, 512, 512), torch.float32), ((6, 3, 1024, 1024), torch.float32),
                                         ((12, 3, 512, 512), torch.float16)])
def test_near_sym_b_scatlayer_on_the_streaming_kernels(shape, dtype):
    import _nearsymb_cases as NB
    NB.check_scat_near_sym_b(DEV, shape, dtype)


@pytest.mark.parametrize('shape,dtype,J,qshift', [((12, 3, 512, 512), torch.float32, 3, 'qshift_d'), ((22, 3, 256, 256), torch.float32, 2, 'qshift_d'),
                                                  ((6, 3, 1024, 1024), torch.float32, 3, 'qshift_d'), ((12, 3, 512, 512), torch.float16, 2, 'qshift_d')])
def test_qshift_d_on_the_streaming_level2_inverse(shape, dtype, J, qshift):
    """Round 6: the 18-tap q-shift filters on the streaming level >= 2 inverse (WlDtInv2Strip<T, 18>; the forward of 18 taps measured
    no faster on the lean kernel and stays on the tile kernel): near_sym_b / qshift_d pyramids against the oracle."""
    import _nearsymb_cases as NB
    kf, ki, kb, _ = NB.check_dtcwt_near_sym_b(DEV, shape, dtype, J=J, qshift=qshift)
    assert any('WlDtInv2Strip<' in k and NB._args(k)[1] == '18' for k in ki), ki


@pytest.mark.parametrize('shape,dtype', [((22, 3, 256, 256), torch.float32), ((12, 3, 512, 512), torch.float32), ((6, 3, 1024, 1024), torch.float32),
                                         ((32, 3, 200, 328), torch.float32), ((12, 3, 512, 512), torch.float16)])
def test_rotationally_symmetric_scatlayer_on_the_lean_kernel(shape, dtype):
    """Round 6: ScatLayer(biort='near_sym_b_bp') inference on the lean streaming kernel (MODE 6) against the oracle and the tile kernel."""
    import _nearsymb_cases as NB
    NB.check_scat_rot_lean(DEV, shape, dtype)


@pytest.mark.parametrize('shape,dtype,stream', [((22, 3, 256, 256), torch.float32, True), ((12, 3, 512, 512), torch.float32, True), ((5, 3, 64, 72), torch.float32, False),
                                                ((12, 3, 512, 512), torch.float16, True)])
def test_rotationally_symmetric_scatlayer_training_step(shape, dtype, stream):
    """Round 6: the training step of ScatLayer(biort='near_sym_b_bp') on two launches of the fused ScatLayer kernels per direction."""
    import _nearsymb_cases as NB
    NB.check_scat_rot_training(DEV, shape, dtype, expect_stream=stream)


@pytest.mark.parametrize('shape,dtype', [((16, 3, 256, 256), torch.float32), ((8, 3, 512, 512), torch.float32), ((4, 3, 72, 88), torch.float32)])
def test_rotationally_symmetric_scatlayerj2_second_order_through_the_layer(shape, dtype):
    """Round 6: ScatLayerj2 with the band-pass tables - the second-order block on the first-order layer's own launches against the chain."""
    import _nearsymb_cases as NB
    ks = NB.check_scatj2_rot(DEV, shape, dtype)
    if shape[-1] >= 256:
        assert any('WlDtFwd12Strip<' in k and NB._args(k)[4] == '6' for k in ks), ks


@pytest.mark.parametrize('shape', [(8, 3, 256, 256), (3, 3, 72, 88)])
def test_rotationally_symmetric_layers_with_colour_combination(shape):
    import _nearsymb_cases as NB
    NB.check_rot_combine_colour(DEV, shape, torch.float32)
