"""The 13 / 19-tap level-1 pair (`near_sym_b`) on the streaming level-1 kernels - the lean forward `WlDtFwd12Strip<T, 13, 19, ..>`
(MODE 0 / 1 / 3) and the streaming inverse `WlDtInv1Strip<T, 19, 13>` (and `<T, 13, 19>` as the forward's backward, `SCAT = 1` as
ScatLayer's) - against the ORACLE (reference dtcwt/transform2d.py:87-147, 193-254; scatternet/lowlevel.py:71-137).  Run by the
emulator (CPU) and the GPU test modules."""
import numpy as np
import torch

import pytorch_wavelets_amd as pw
from oracle import wavelet_oracle as wo
from pytorch_wavelets_amd import filters as F


def _args(k):
    return [a.strip() for a in k[k.index('<') + 1:k.rindex('>')].split(',')]


def _has(ks, name, *taps):
    return any((name + '<') in k and _args(k)[1:1 + len(taps)] == [str(t) for t in taps] for k in ks)


def _npy(t):
    return t.detach().cpu().double().numpy()


def _rel(a, b):
    return float(np.abs(a - b).max()) / float(np.abs(b).max())


def check_dtcwt_near_sym_b(dev, shape, dtype, J=1, qshift='qshift_b', tol=1e-5, expect_stream=True):
    """DTCWTForward / DTCWTInverse / the forward's gradient with `near_sym_b`: the kernels named, every output against the oracle."""
    rng = np.random.RandomState(7)
    x = rng.randn(*shape)
    if dtype == torch.float16:
        x = np.float16(x).astype(np.float64)
    fw = F.dtcwt_forward_taps('near_sym_b', qshift)
    iv = F.dtcwt_inverse_taps('near_sym_b', qshift)
    assert len(fw[0]) == 13 and len(fw[1]) == 19 and len(iv[0]) == 19 and len(iv[1]) == 13
    want_l, want_h = wo.dtcwt_forward(x, J, *fw)
    xfm = pw.DTCWTForward(J=J, biort='near_sym_b', qshift=qshift).to(dev).to(dtype)
    ifm = pw.DTCWTInverse(biort='near_sym_b', qshift=qshift).to(dev).to(dtype)
    xt = torch.tensor(x, dtype=dtype, device=dev).requires_grad_(dtype != torch.float16)
    c0 = pw.launch_count()
    yl, yh = xfm(xt)
    kf = pw.kernels_since(c0)
    t = 4e-3 if dtype == torch.float16 else tol
    assert _rel(_npy(yl), want_l) <= t, (shape, kf)
    for a, b in zip(yh, want_h):
        assert a.shape == b.shape and _rel(_npy(a), b) <= t, (shape, kf)
    # inverse of perturbed coefficients (not a round trip: the bands carry independent data)
    cl = want_l + 0.1 * rng.randn(*want_l.shape)
    ch = [h + 0.1 * rng.randn(*h.shape) for h in want_h]
    if dtype == torch.float16:
        cl, ch = np.float16(cl).astype(np.float64), [np.float16(h).astype(np.float64) for h in ch]
    want_y = wo.dtcwt_inverse(cl, ch, *iv)
    c0 = pw.launch_count()
    y = ifm((torch.tensor(cl, dtype=dtype, device=dev), [torch.tensor(h, dtype=dtype, device=dev) for h in ch]))
    ki = pw.kernels_since(c0)
    assert y.shape == want_y.shape and _rel(_npy(y), want_y) <= (1e-2 if dtype == torch.float16 else tol), (shape, ki)
    kb = []
    if dtype != torch.float16:
        # the gradient is the adjoint: <dx, v> = <cotangent, forward(v)> for any v (the oracle's forward, float64)
        cot_l = rng.randn(*want_l.shape)
        cot_h = [rng.randn(*h.shape) for h in want_h]
        c0 = pw.launch_count()
        dx, = torch.autograd.grad((yl * torch.tensor(cot_l, dtype=dtype, device=dev)).sum()
                                  + sum((a * torch.tensor(b, dtype=dtype, device=dev)).sum() for a, b in zip(yh, cot_h)), xt)
        kb = pw.kernels_since(c0)
        dxn = _npy(dx)
        for _ in range(3):
            v = rng.randn(*shape)
            vl, vh = wo.dtcwt_forward(v, J, *fw)
            rhs = float((cot_l * vl).sum() + sum((a * b).sum() for a, b in zip(cot_h, vh)))
            lhs = float((dxn * v).sum())
            scale = float(np.sqrt((dxn ** 2).sum() * (v ** 2).sum()))
            assert abs(lhs - rhs) <= 2e-5 * scale, (shape, lhs, rhs, kb)
    if expect_stream:
        assert _has(kf, 'WlDtFwd12Strip', 13, 19), kf
        assert _has(ki, 'WlDtInv1Strip', 19, 13), ki
        if kb:
            assert _has(kb, 'WlDtInv1Strip', 13, 19), kb
    return kf, ki, kb, dxn if dtype != torch.float16 else None


def check_scat_near_sym_b(dev, shape, dtype, tol=1e-5, expect_stream=True):
    """ScatLayer(biort='near_sym_b'): inference and training forward, and the backward, against the oracle."""
    rng = np.random.RandomState(11)
    x = rng.randn(*shape)
    if dtype == torch.float16:
        x = np.float16(x).astype(np.float64)
    h0o, h1o = F.dtcwt_forward_taps('near_sym_b', 'qshift_a')[:2]
    Z, saved = wo.scat_layer_forward(x, h0o, h1o, return_saved=True)
    dZ = rng.randn(*Z.shape)
    want = wo.scat_layer_backward(dZ, saved, h0o, h1o)
    sl = pw.ScatLayer(biort='near_sym_b').to(dev).to(dtype)
    t = 1e-2 if dtype == torch.float16 else tol
    with torch.no_grad():
        c0 = pw.launch_count()
        z0 = sl(torch.tensor(x, dtype=dtype, device=dev))
        k0 = pw.kernels_since(c0)
    assert _rel(_npy(z0), Z) <= t, (shape, k0)
    xg = torch.tensor(x, dtype=dtype, device=dev).requires_grad_(True)
    c0 = pw.launch_count()
    z = sl(xg)
    k1 = pw.kernels_since(c0)
    assert _rel(_npy(z), Z) <= t, (shape, k1)
    c0 = pw.launch_count()
    g, = torch.autograd.grad(z, xg, torch.tensor(dZ, dtype=dtype, device=dev))
    kb = pw.kernels_since(c0)
    assert g.shape == want.shape and _rel(_npy(g), want) <= t, (shape, kb)
    if expect_stream:
        assert any('WlDtFwd12Strip<' in k and _args(k)[1:3] == ['13', '19'] and _args(k)[4] == '1' for k in k0), k0
        assert any('WlDtFwd12Strip<' in k and _args(k)[1:3] == ['13', '19'] and _args(k)[4] == '3' for k in k1), k1
        assert any('WlDtInv1Strip<' in k and _args(k)[1:4] == ['13', '19', '1'] for k in kb), kb
    return k0, k1, kb


def check_scat_rot_lean(dev, shape, dtype, tol=1e-5, expect_stream=True):
    """ScatLayer(biort='near_sym_b_bp') inference on the lean kernel with the third (band-pass) row filter and window
    (WlDtFwd12Strip<T, 13, 19, 10, 6, ..>) against the oracle's single-axis filters (fwd_j1_rot + the magnitudes of
    ScatLayerj1_rot_f.forward, scatternet/lowlevel.py:140-182) and, elementwise, against the tile kernel WlDtFwd1Rot."""
    import _ext_cases as E
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dtcwt import lowlevel as dl
    rng = np.random.RandomState(13)
    x = rng.randn(*shape)
    if dtype == torch.float16:
        x = np.float16(x).astype(np.float64)
    h0o, _, h1o, _, h2o, _ = F.biort('near_sym_b_bp')
    hn = [dl.prep_filt(v, 1).to(torch.float64).numpy().ravel() for v in (h0o, h1o, h2o)]
    ll, re, im = E.rot_level1_reference(x, *hn, 'symmetric')
    pool = ll.reshape(ll.shape[0], ll.shape[1], ll.shape[2] // 2, 2, ll.shape[3] // 2, 2).mean(axis=(3, 5))
    want = np.concatenate([pool[:, None], np.sqrt(re ** 2 + im ** 2 + 1e-4) - 0.01], 1)
    sl = pw.ScatLayer(biort='near_sym_b_bp').to(dev).to(dtype)
    xt = torch.tensor(x, dtype=dtype, device=dev)
    t = 4e-3 if dtype == torch.float16 else tol
    with torch.no_grad():
        c0 = pw.launch_count()
        z = sl(xt)
        k0 = pw.kernels_since(c0)
        try:
            ops.set_option('no_stream', 1)
            c0 = pw.launch_count()
            z1 = sl(xt)
            k1 = pw.kernels_since(c0)
        finally:
            ops.set_option('no_stream', 0)
    n, _, c, h, w = want.shape
    assert z.shape == (n, 7 * c, h, w)
    assert _rel(_npy(z).reshape(want.shape), want) <= t, (shape, k0)
    assert _rel(_npy(z1).reshape(want.shape), want) <= t, (shape, k1)
    assert float((z.float() - z1.float()).abs().max()) <= (t if dtype == torch.float16 else 3e-6) * float(z1.float().abs().max())
    assert all(k.startswith('WlDtFwd1Rot<') for k in k1), k1
    if expect_stream:
        assert len(k0) == 1 and 'WlDtFwd12Strip<' in k0[0] and _args(k0[0])[1:3] == ['13', '19'] and _args(k0[0])[4] == '6', k0
    return k0


def check_scat_rot_training(dev, shape, dtype, tol=2e-5, expect_stream=True):
    """The training step of ScatLayer(biort='near_sym_b_bp') on two launches of the fused ScatLayer kernels per direction
    (scatternet/lowlevel.py ScatLayerj1_rot_train_f) against the chain of differentiable pieces it replaces (FWD_J1_ROT + the
    tensor library's magnitudes; pinned to the reference's Z and dx by the goldens ext_rot_*) and against the oracle's Z."""
    import _ext_cases as E
    from pytorch_wavelets_amd.dtcwt import lowlevel as dl
    from pytorch_wavelets_amd.scatternet import lowlevel as sl_ll
    rng = np.random.RandomState(17)
    x = rng.randn(*shape)
    if dtype == torch.float16:
        x = np.float16(x).astype(np.float64)
    h0o, _, h1o, _, h2o, _ = F.biort('near_sym_b_bp')
    hn = [dl.prep_filt(v, 1).to(torch.float64).numpy().ravel() for v in (h0o, h1o, h2o)]
    ll, re, im = E.rot_level1_reference(x, *hn, 'symmetric')
    pool = ll.reshape(ll.shape[0], ll.shape[1], ll.shape[2] // 2, 2, ll.shape[3] // 2, 2).mean(axis=(3, 5))
    want = np.concatenate([pool[:, None], np.sqrt(re ** 2 + im ** 2 + 1e-4) - 0.01], 1)
    sl = pw.ScatLayer(biort='near_sym_b_bp').to(dev).to(dtype)
    gz = torch.tensor(rng.randn(want.shape[0], 7 * want.shape[2], *want.shape[3:]), dtype=dtype, device=dev)
    out = {}
    for fused in (True, False):
        sl_ll.ROT_TRAIN_FUSED = fused
        try:
            xg = torch.tensor(x, dtype=dtype, device=dev).requires_grad_(True)
            c0 = pw.launch_count()
            z = sl(xg)
            g, = torch.autograd.grad(z, xg, gz)
            out[fused] = (z.detach(), g, pw.kernels_since(c0))
        finally:
            sl_ll.ROT_TRAIN_FUSED = True
    t = 1e-2 if dtype == torch.float16 else tol
    assert _rel(_npy(out[True][0]).reshape(want.shape), want) <= (4e-3 if dtype == torch.float16 else 1e-5), shape
    for a, b in zip(out[True][:2], out[False][:2]):
        assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= t * float(b.float().abs().max()), (shape, out[True][2])
    ks = out[True][2]
    assert len(ks) == 4, ks
    if expect_stream:
        assert sum('WlDtFwd12Strip<' in k and _args(k)[1:3] == ['13', '19'] and _args(k)[4] == '3' for k in ks) == 2, ks
        assert sum('WlDtInv1Strip<' in k and _args(k)[1:4] == ['13', '19', '1'] for k in ks) == 2, ks
    return ks


def check_scatj2_rot(dev, shape, dtype, tol=3e-5):
    """ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_b_bp'): its second-order block through the first-order band-pass layer's own
    launches (inference: lean MODE 6 / WlDtFwd1Rot; training: ScatLayerj1_rot_train_f) against the chain of round 4 (pinned to the
    reference's Z and dx by the goldens ext_rot_3 / ext_rot_4) - output and input gradient."""
    from pytorch_wavelets_amd.scatternet import lowlevel as sl_ll
    torch.manual_seed(19)
    x = torch.randn(*shape, device=dev).to(dtype)
    sl = pw.ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_b_bp').to(dev).to(dtype)
    out = {}
    for fused in (True, False):
        sl_ll.ROT_TRAIN_FUSED = fused
        try:
            with torch.no_grad():
                c0 = pw.launch_count()
                z0 = sl(x)
                k0 = pw.kernels_since(c0)
            xg = x.clone().requires_grad_(True)
            z = sl(xg)
            g, = torch.autograd.grad(z, xg, torch.ones_like(z) + 0.5 * z.detach())
            out[fused] = (z0, z.detach(), g, k0)
        finally:
            sl_ll.ROT_TRAIN_FUSED = True
    t = 1e-2 if dtype == torch.float16 else tol
    for a, b in zip(out[True][:3], out[False][:3]):
        assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= t * float(b.float().abs().max()), (shape, out[True][3])
    return out[True][3]


def check_rot_combine_colour(dev, shape, dtype, tol=3e-5):
    """combine_colour=True with the band-pass tables (ScatLayer and ScatLayerj2): the pairs of plain launches against the chain of round 4
    (pinned to the reference by the goldens ext_rot_1 / ext_rot_4) - output and input gradient."""
    from pytorch_wavelets_amd.scatternet import lowlevel as sl_ll
    torch.manual_seed(23)
    x = torch.randn(*shape, device=dev).to(dtype)
    for mod in (pw.ScatLayer(biort='near_sym_b_bp', combine_colour=True), pw.ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_b_bp', combine_colour=True)):
        mod = mod.to(dev).to(dtype)
        out = {}
        for fused in (True, False):
            sl_ll.ROT_TRAIN_FUSED = fused
            try:
                with torch.no_grad():
                    z0 = mod(x)
                xg = x.clone().requires_grad_(True)
                c0 = pw.launch_count()
                z = mod(xg)
                g, = torch.autograd.grad(z, xg, torch.ones_like(z) + 0.5 * z.detach())
                out[fused] = (z0, z.detach(), g, pw.kernels_since(c0))
            finally:
                sl_ll.ROT_TRAIN_FUSED = True
        assert not any('WlCorr1d' in k or 'WlDtFwd1Rot' in k for k in out[True][3]), out[True][3]
        assert any('WlCorr1d' in k for k in out[False][3]), out[False][3]
        for a, b in zip(out[True][:3], out[False][:3]):
            assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max()), (shape, type(mod).__name__)
