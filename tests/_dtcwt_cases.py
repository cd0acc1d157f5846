"""Shared DTCWT / ScatLayer golden checks, run by the emulator (CPU) and the GPU test modules."""
import numpy as np
import torch

import _golden as G
import pytorch_wavelets_amd as pw


def check_dtcwt_case(name, dev, dtype, tol, grads=True):
    meta, g = G.INDEX[name], G.load(name)
    npdt = np.float64 if dtype == torch.float64 else np.float32

    def t(a):
        return torch.tensor(np.asarray(a, dtype=npdt), device=dev)

    def npy(v):
        return v.detach().cpu().double().numpy()

    xfm = pw.DTCWTForward(biort=meta['biort'], qshift=meta['qshift'], J=meta['J'], skip_hps=meta['skip_hps'],
                          include_scale=meta['include_scale'], mode=meta['mode']).to(dev)
    ifm = pw.DTCWTInverse(biort=meta['biort'], qshift=meta['qshift'], mode=meta['mode']).to(dev)
    x = t(g['x']).requires_grad_(True)
    yl, yh = xfm(x)
    if isinstance(yl, list):
        for j, s in enumerate(yl):
            if s.shape != torch.Size([]):
                assert G.relerr(npy(s), g, 'scale%d' % j) < tol
        low = [s for s in yl if s.shape != torch.Size([])][-1]
    else:
        low = yl
    assert G.relerr(npy(low), g, 'yl') < tol
    for j, h in enumerate(yh):
        if h.shape == torch.Size([]):
            assert not G.has(g, 'yh%d' % j)
        else:
            assert h.is_contiguous()
            assert G.relerr(npy(h), g, 'yh%d' % j) < tol
    rec = ifm((low, yh))
    assert G.relerr(npy(rec), g, 'rec') < tol
    if not grads:
        return
    if 'dx' in g:
        outs = [low] + [h for h in yh if h.shape != torch.Size([])]
        dx, = torch.autograd.grad(sum((o * t(g['cot%d' % i])).sum() for i, o in enumerate(outs)), x)
        assert G.relerr(npy(dx), g, 'dx') < tol
    if 'dinv0' in g:
        lowr = t(g['yl']).requires_grad_(True)
        yhr = [t(g['yh%d' % j]).requires_grad_(True) if G.has(g, 'yh%d' % j) else torch.zeros([], device=dev)
               for j in range(meta['J'])]
        gin = [lowr] + [h for h in yhr if h.requires_grad]
        gr = torch.autograd.grad((ifm((lowr, yhr)) * t(g['gy'])).sum(), gin)
        for i, a in enumerate(gr):
            assert G.relerr(npy(a), g, 'dinv%d' % i) < tol


def check_dtcwt_none(dev, dtype, tol):
    g = G.load('dtcwt_none')
    npdt = np.float64 if dtype == torch.float64 else np.float32

    def t(a):
        return torch.tensor(np.asarray(a, dtype=npdt), device=dev)
    ifm = pw.DTCWTInverse().to(dev)
    yl, y0, y1, y2 = t(g['yl']), t(g['yh0']), t(g['yh1']), t(g['yh2'])
    assert G.relerr(ifm((yl, [None, y1, y2])).cpu().double().numpy(), g, 'rec_a') < tol
    assert G.relerr(ifm((yl, [y0, torch.zeros([], device=dev), y2])).cpu().double().numpy(), g, 'rec_b') < tol
    assert G.relerr(ifm((torch.zeros_like(yl), [y0, y1, y2])).cpu().double().numpy(), g, 'rec_c') < tol


def check_scat_case(name, dev, dtype, tol):
    meta, g = G.INDEX[name], G.load(name)
    npdt = np.float64 if dtype == torch.float64 else np.float32

    def t(a):
        return torch.tensor(np.asarray(a, dtype=npdt), device=dev)
    sl = pw.ScatLayer(biort=meta['biort'], mode=meta['mode'], magbias=meta['magbias'],
                      combine_colour=meta['combine_colour']).to(dev)
    x = t(g['x']).requires_grad_(True)
    Z = sl(x)
    assert G.relerr(Z.detach().cpu().double().numpy(), g, 'Z') < tol
    if 'dx' in g:
        dx, = torch.autograd.grad((Z * t(g['gz'])).sum(), x)
        assert G.relerr(dx.cpu().double().numpy(), g, 'dx') < tol


def check_layouts(dev, dtype, tol):
    """o_dim / ri_dim permutations (reference tests/test_dtcwt.py:188-214, :297-319)."""
    torch.manual_seed(7)
    x = torch.randn(2, 3, 32, 24, dtype=dtype, device=dev)
    ref_yl, ref_yh = pw.DTCWTForward(J=2).to(dev)(x)
    for o_dim, ri_dim in ((1, -1), (2, 3), (4, 1), (3, 2), (5, 1), (1, 2)):
        xfm = pw.DTCWTForward(J=2, o_dim=o_dim, ri_dim=ri_dim).to(dev)
        ifm = pw.DTCWTInverse(o_dim=o_dim, ri_dim=ri_dim).to(dev)
        yl, yh = xfm(x)
        for a, b in zip(yh, ref_yh):
            assert a.shape[o_dim] == 6 and a.shape[ri_dim] == 2 and a.is_contiguous()
            # move back to the default layout and compare
            rest = [d for d in range(6) if d not in (o_dim % 6, ri_dim % 6)]
            back = a.permute(rest[0], rest[1], o_dim % 6, rest[2], rest[3], ri_dim % 6)
            assert torch.equal(back, b)
        rec = ifm((yl, yh))
        assert float((rec - x).abs().max() / x.abs().max()) < tol


def check_scat_backward_streaming(dev, shapes, tol=1e-5):
    """ScatLayer's backward on the streaming level-1 inverse with the scattering prologue in its stagers
    (WlDtInv1Strip<..., SCAT = 1>) against the ORACLE's ScatLayerj1_f.backward (forward, saved quotients and gradient all from
    the oracle in float64): whole planes, several strips / segments, float16."""
    from oracle import wavelet_oracle as wo
    from pytorch_wavelets_amd import filters as F
    rng = np.random.RandomState(41)
    h0o, h1o = F.dtcwt_forward_taps('near_sym_a', 'qshift_a')[:2]
    for shape, dtype in shapes:
        x = rng.randn(*shape)
        if dtype == torch.float16:
            x = np.float16(x).astype(np.float64)
        Z, saved = wo.scat_layer_forward(x, h0o, h1o, return_saved=True)
        dZ = rng.randn(*Z.shape)
        want = wo.scat_layer_backward(dZ, saved, h0o, h1o)
        sl = pw.ScatLayer().to(dev).to(dtype)
        xg = torch.tensor(x, dtype=dtype, device=dev).requires_grad_(True)
        z = sl(xg)
        t = 1e-2 if dtype == torch.float16 else tol   # (float16: dZ, the saved quotients and the result are each rounded to 11 bits; the maximum over 10^7 samples)
        assert float(np.abs(z.detach().cpu().double().numpy() - Z).max()) <= t * float(np.abs(Z).max())
        c0 = pw.launch_count()
        g, = torch.autograd.grad(z, xg, torch.tensor(dZ, dtype=dtype, device=dev))
        ks = pw.kernels_since(c0)
        # <T, L0, L1, SCAT = 1 (, PP = 2 for planes of up to 256 columns)>
        assert any('WlDtInv1Strip<' in k and [a.strip() for a in k[k.index('<') + 1:k.rindex('>')].split(',')][3:4] == ['1'] for k in ks), ks
        assert g.shape == want.shape
        assert float(np.abs(g.detach().cpu().double().numpy() - want).max()) <= t * float(np.abs(want).max()), (shape, dtype)
