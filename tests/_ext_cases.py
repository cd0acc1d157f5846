"""Shared checks for the single-axis APIs (1-D DWT, stationary transform, DTCWT primitives, function-level banks) against
the goldens generated from the real reference by oracle/pin_extras.py.  Used by the emulator tests (CPU tensors) and by
the -m gpu tests (device tensors, the real library)."""
import numpy as np
import torch

import _golden as G
from oracle import wavelet_oracle as wo
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import filters
from pytorch_wavelets_amd.dtcwt import lowlevel as dtl
from pytorch_wavelets_amd.dwt import lowlevel as dwl
from pytorch_wavelets_amd.dwt.transform2d import SWTForward


def _t(a, dev, dtype):
    return torch.tensor(np.asarray(a), device=dev).to(dtype)


def check_dwt1d(name, dev, dtype, tol):
    meta, g = G.INDEX[name], G.load(name)
    J = meta['J']
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        xfm = pw.DWT1DForward(J=J, wave=meta['wave'], mode=meta['mode']).to(dev)
        ifm = pw.DWT1DInverse(wave=meta['wave'], mode=meta['mode']).to(dev)
    finally:
        torch.set_default_dtype(prev)
    assert xfm.h0.shape == (1, 1, len(filters.Wavelet(meta['wave']).dec_lo))
    x = _t(g['x'], dev, dtype).requires_grad_(True)
    yl, yh = xfm(x)
    assert G.relerr(yl.detach().cpu().numpy(), g, 'yl') < tol
    for j in range(J):
        assert G.relerr(yh[j].detach().cpu().numpy(), g, 'yh%d' % j) < tol
    rec = ifm((yl, yh))
    assert G.relerr(rec.detach().cpu().numpy(), g, 'rec') < tol
    loss = (yl * _t(g['gl'], dev, dtype)).sum() + sum((yh[j] * _t(g['gh%d' % j], dev, dtype)).sum() for j in range(J))
    dx, = torch.autograd.grad(loss, x)
    assert G.relerr(dx.cpu().numpy(), g, 'dx') < tol
    ylr = _t(g['yl'], dev, dtype).requires_grad_(True)
    yhr = [_t(g['yh%d' % j], dev, dtype).requires_grad_(True) for j in range(J)]
    gr = torch.autograd.grad((ifm((ylr, yhr)) * _t(g['gy'], dev, dtype)).sum(), [ylr] + yhr)
    assert G.relerr(gr[0].cpu().numpy(), g, 'dyl') < tol
    for j in range(J):
        assert G.relerr(gr[1 + j].cpu().numpy(), g, 'dyh%d' % j) < tol
    # None highs are zeros (transform1d.py:104-106)
    rec0 = ifm((yl.detach(), [None] * J))   # (no 'unpad' happens then, exactly as upstream)
    assert rec0.ndim == 3 and bool(torch.isfinite(rec0).all())


def check_swt(name, dev, dtype, tol):
    meta, g = G.INDEX[name], G.load(name)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        m = SWTForward(J=2, wave=meta['wave'], mode=meta['mode']).to(dev)
    finally:
        torch.set_default_dtype(prev)
    x = _t(g['x'], dev, dtype)
    y = m(x)
    assert len(y) == 2 and y[0].shape == y[1].shape == (x.shape[0], 4 * x.shape[1]) + tuple(x.shape[2:])
    assert G.relerr(y[0].cpu().numpy(), g, 'y') < tol
    # the dilated bank on the same input (the level-2 OPERATOR of the reference; its SWTForward cannot reach level 2)
    filts = (m.h0_col, m.h1_col, m.h0_row, m.h1_row)
    assert G.relerr(dwl.afb2d_atrous(x, filts, meta['mode'], 2).cpu().numpy(), g, 'y_dil2') < tol


def check_prims(dev, dtype, tol):
    g = G.load('ext_prims')
    h0o, g0o, h1o, g1o = filters.biort('near_sym_b')
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = filters.qshift('qshift_b')
    P = lambda v: dtl.prep_filt(v, 1).to(dev).to(dtype)   # noqa: E731
    X = _t(g['X'], dev, dtype)
    res = {
        'colfilter': dtl.colfilter(X, P(h1o)), 'rowfilter': dtl.rowfilter(X, P(h0o)),
        'colfilter_zero': dtl.colfilter(X, P(h0o), 'zero'),
        'coldfilt': dtl.coldfilt(X, P(h0b), P(h0a)), 'coldfilt_hp': dtl.coldfilt(X, P(h1b), P(h1a), True),
        'rowdfilt': dtl.rowdfilt(X, P(h0b), P(h0a)), 'rowdfilt_hp': dtl.rowdfilt(X, P(h1b), P(h1a), True),
        'colifilt': dtl.colifilt(X, P(g0b), P(g0a)), 'colifilt_hp': dtl.colifilt(X, P(g1b), P(g1a), True),
        'rowifilt': dtl.rowifilt(X, P(g0b), P(g0a)), 'rowifilt_hp': dtl.rowifilt(X, P(g1b), P(g1a), True),
    }
    for k, v in res.items():
        assert G.relerr(v.cpu().numpy(), g, k) < tol, k
    (a, b), (c, d) = dtl.q2c(X)
    for k, v in (('q2c_1r', a), ('q2c_1i', b), ('q2c_2r', c), ('q2c_2i', d), ('c2q', dtl.c2q((a, b), (c, d)))):
        assert G.relerr(v.cpu().numpy(), g, k) < tol, k
    import pytest
    with pytest.raises(ValueError, match='multiple of 4'):
        dtl.coldfilt(X[:, :, :14], P(h0b), P(h0a))
    with pytest.raises(ValueError, match='multiple of 2'):
        dtl.rowifilt(X[:, :, :, :23], P(g0b), P(g0a))


def check_afb1d_functions(dev, tol):
    g = G.load('ext_afb1d')
    w = filters.Wavelet('db3')
    x = torch.tensor(g['x'], device=dev)
    lohi = dwl.afb1d(x, w.dec_lo, w.dec_hi, mode='symmetric', dim=3)
    assert G.relerr(lohi.cpu().numpy(), g, 'lohi') < tol
    y = dwl.sfb1d(lohi[:, ::2].contiguous(), lohi[:, 1::2].contiguous(), w.rec_lo, w.rec_hi, mode='symmetric', dim=3)
    assert G.relerr(y.cpu().numpy(), g, 'y') < tol


def check_afb1d_periodization(dev, dtype, tol):
    """Function-level afb1d / sfb1d in mode 'periodization' with odd tap counts and signals shorter than the filter (the
    reference rolls, convolves with zero padding and folds the wrapped tail ONCE, dwt/lowlevel.py:134-150, :252-261)."""
    meta, g = G.INDEX['ext_afb1d_per'], G.load('ext_afb1d_per')
    for c in meta['cases']:
        k, d = c['key'], c['dim']
        x = _t(g[k + '_x'], dev, dtype)
        h0, h1 = _t(g[k + '_h0'], dev, dtype), _t(g[k + '_h1'], dev, dtype)   # tensors: taken as already reversed
        lohi = dwl.afb1d(x, h0, h1, mode='periodization', dim=d)
        assert G.relerr(lohi.cpu().numpy(), g, k + '_lohi') < tol, c
        if c.get('syn'):
            lo, hi = lohi[:, ::2].contiguous(), lohi[:, 1::2].contiguous()
            y = dwl.sfb1d(lo, hi, _t(g[k + '_g0'], dev, dtype), _t(g[k + '_g1'], dev, dtype), mode='periodization', dim=d)
            assert G.relerr(y.cpu().numpy(), g, k + '_y') < tol, c


def check_scatj2(name, dev, dtype, tol):
    meta, g = G.INDEX[name], G.load(name)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        m = pw.ScatLayerj2(combine_colour=meta['combine_colour']).to(dev)
    finally:
        torch.set_default_dtype(prev)
    assert sorted(n for n, _ in m.named_parameters()) == ['h0a', 'h0b', 'h0o', 'h1a', 'h1b', 'h1o']
    x = _t(g['x'], dev, dtype).requires_grad_(True)
    Z = m(x)
    assert G.relerr(Z.detach().cpu().numpy(), g, 'Z') < tol
    dx, = torch.autograd.grad((Z * _t(g['gz'], dev, dtype)).sum(), x)
    assert G.relerr(dx.cpu().numpy(), g, 'dx') < tol
    with torch.no_grad():
        assert G.relerr(m(x.detach()).cpu().numpy(), g, 'Z') < tol   # no-grad path (nothing saved)


def check_rot(name, dev, dtype, tol):
    meta, g = G.INDEX[name], G.load(name)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        m = getattr(pw, meta['cls'])(**meta['kwargs']).to(dev)
    finally:
        torch.set_default_dtype(prev)
    assert m.bandpass_diag and 'h2o' in dict(m.named_parameters())
    x = _t(g['x'], dev, dtype).requires_grad_(True)
    Z = m(x)
    assert G.relerr(Z.detach().cpu().numpy(), g, 'Z') < tol
    dx, = torch.autograd.grad((Z * _t(g['gz'], dev, dtype)).sum(), x)
    assert G.relerr(dx.cpu().numpy(), g, 'dx') < tol
    with torch.no_grad():
        assert G.relerr(m(x.detach()).cpu().numpy(), g, 'Z') < tol   # no-grad path (ScatLayer: magnitudes from the same launch)


def rot_level1_reference(x, h0, h1, h2, mode):
    """fwd_j1_rot (reference dtcwt/transform_funcs.py:124-149) from the oracle's single-axis filters: ll (N,C,H,W) and the six
    orientations (N,6,C,H/2,W/2) as (real, imaginary)."""
    lo, hi, ba = (wo.rowfilter(x, h, mode) for h in (h0, h1, h2))
    ll, lh, hl, hh = wo.colfilter(lo, h0, mode), wo.colfilter(lo, h1, mode), wo.colfilter(hi, h0, mode), wo.colfilter(ba, h2, mode)

    def q2c(y):
        y = y / np.sqrt(2)
        a, b, c, d = y[:, :, 0::2, 0::2], y[:, :, 0::2, 1::2], y[:, :, 1::2, 0::2], y[:, :, 1::2, 1::2]
        return (a - d, b + c), (a + d, b - c)
    (r15, i15), (r165, i165) = q2c(lh)
    (r45, i45), (r135, i135) = q2c(hh)
    (r75, i75), (r105, i105) = q2c(hl)
    return ll, np.stack([r15, r45, r75, r105, r135, r165], 1), np.stack([i15, i45, i75, i105, i135, i165], 1)


DWT1D_CASES = sorted(k for k, v in G.INDEX.items() if v['kind'] == 'dwt1d')
SWT_CASES = sorted(k for k, v in G.INDEX.items() if v['kind'] == 'swt')
SCATJ2_CASES = sorted(k for k, v in G.INDEX.items() if v['kind'] == 'scatj2')
ROT_CASES = sorted(k for k, v in G.INDEX.items() if v['kind'] == 'rot')


NONSEP_CASES = sorted(k for k, v in G.INDEX.items() if v['kind'] == 'nonsep')


def check_nonsep(name, dev, dtype, tol):
    """afb2d_nonsep / sfb2d_nonsep (+ their prep_filt functions) against the reference's outputs and against the
    gradients autograd gives upstream (every mode)."""
    import pytest
    meta, g = G.INDEX[name], G.load(name)
    mode = meta['mode']
    fa, fs = _t(g['fa'], dev, dtype), _t(g['fs'], dev, dtype)
    wcol, wrow = meta['wave']
    if wcol != 'random':   # the prep functions build the same point-spread functions as upstream
        wc = filters.Wavelet(wcol)
        wr = filters.Wavelet(wrow) if wrow else wc
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            pa = dwl.prep_filt_afb2d_nonsep(wc.dec_lo, wc.dec_hi, wr.dec_lo, wr.dec_hi)
            ps = dwl.prep_filt_sfb2d_nonsep(wc.rec_lo, wc.rec_hi, wr.rec_lo, wr.rec_hi)
        finally:
            torch.set_default_dtype(prev)
        assert pa.shape == fa.shape and np.abs(pa.numpy() - g['fa']).max() < 1e-6
        assert ps.shape == fs.shape and np.abs(ps.numpy() - g['fs']).max() < 1e-6
    if mode == 'periodic':
        with pytest.raises(ValueError, match='Unkown pad type'):
            dwl.afb2d_nonsep(_t(g['x'], dev, dtype), fa, mode)
    else:
        x = _t(g['x'], dev, dtype).requires_grad_(True)
        y = dwl.afb2d_nonsep(x, fa, mode)
        assert G.relerr(y.detach().cpu().numpy(), g, 'y') < tol
        dx, = torch.autograd.grad((y * _t(g['gy'], dev, dtype)).sum(), x)   # every mode: the true adjoint, as upstream
        assert G.relerr(dx.cpu().numpy(), g, 'dx') < tol
    c = _t(g['c'], dev, dtype).requires_grad_(True)
    rec = dwl.sfb2d_nonsep(c, fs, mode)
    assert G.relerr(rec.detach().cpu().numpy(), g, 'rec') < tol
    dc, = torch.autograd.grad((rec * _t(g['gr'], dev, dtype)).sum(), c)
    assert G.relerr(dc.cpu().numpy(), g, 'dc') < tol


def check_dwt1d_fused(dev, cases=None, tol=1e-5):
    """DWT1DForward on the fused multi-level 1-D kernel (csrc/wl_dwt1d_fused.h) against the ORACLE: every mode, odd lengths,
    rows of one chunk and of many chunks (chunk ends inside the signal, at its ends, shorter last chunks), 2-20 taps, J = 1..4,
    float16; and its gradient against the per-level path."""
    import numpy as np
    from oracle import wavelet_oracle as wo
    from pytorch_wavelets_amd import filters as F
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(23)
    cases = cases or [('db4', 'symmetric', 3, (2, 3, 9000), torch.float32), ('db1', 'zero', 4, (1, 2, 12305), torch.float32),
                      ('db3', 'reflect', 2, (3, 1, 4097), torch.float32), ('db8', 'symmetric', 3, (1, 2, 20000), torch.float32),
                      ('sym10', 'zero', 2, (1, 1, 8191), torch.float32), ('db2', 'periodization', 3, (2, 2, 1000), torch.float32),
                      ('db5', 'periodic', 2, (1, 3, 777), torch.float32), ('db6', 'periodization', 1, (2, 1, 333), torch.float32),
                      ('db4', 'symmetric', 1, (4, 4, 64), torch.float32), ('coif2', 'reflect', 4, (1, 1, 16384), torch.float32),
                      ('db4', 'symmetric', 3, (2, 2, 8192), torch.float16), ('db7', 'zero', 3, (1, 2, 5001), torch.float32),
                      # many rows (more workgroups than the emulated chip holds at once)
                      ('db3', 'symmetric', 2, (40, 2, 6000), torch.float32), ('db2', 'zero', 1, (60, 1, 9001), torch.float32)]
    for wave, mode, J, shape, dtype in cases:
        h0, h1 = F.dwt_analysis_taps(wave)
        x = torch.tensor(rng.randn(*shape), dtype=dtype, device=dev)
        oyl, oyh = wo.dwt1d_forward(x.detach().cpu().double().numpy(), J, h0, h1, mode)
        xfm = pw.DWT1DForward(J=J, wave=wave, mode=mode).to(dev).to(dtype)
        xa = x.clone().requires_grad_(dtype == torch.float32)
        yl, yh = xfm(xa)
        assert 'WlDwt1dFused' in pw.last_kernel(), (wave, mode, J, shape, pw.last_kernel())
        t = 4e-3 if dtype == torch.float16 else tol
        assert yl.shape == oyl.shape
        assert np.abs(yl.detach().cpu().double().numpy() - oyl).max() <= t * np.abs(oyl).max(), (wave, mode, J, shape, 'yl')
        for j in range(J):
            assert yh[j].shape == oyh[j].shape and yh[j].is_contiguous()
            assert np.abs(yh[j].detach().cpu().double().numpy() - oyh[j]).max() <= t * max(np.abs(oyh[j]).max(), 1e-30), (wave, mode, J, shape, j)
        # the inverse on the fused multi-level 1-D synthesis kernel (csrc/wl_idwt1d_fused.h): against the oracle
        g0, g1 = F.dwt_synthesis_taps(wave)
        ifm = pw.DWT1DInverse(wave=wave, mode=mode).to(dev).to(dtype)
        c0 = pw.launch_count()
        rec = ifm((yl.detach(), [h.detach() for h in yh]))
        if mode != 'periodization':
            assert pw.kernels_since(c0)[0].startswith('WlIdwt1dFused'), (wave, mode, J, shape, pw.kernels_since(c0))
        orec = wo.dwt1d_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh], g0, g1, mode)
        assert rec.shape == orec.shape
        assert np.abs(rec.cpu().double().numpy() - orec).max() <= (6e-3 if dtype == torch.float16 else 2 * tol) * max(1.0, np.abs(orec).max()), (wave, mode, J, shape, 'rec')
        if dtype == torch.float32:
            ws = [torch.randn_like(yl)] + [torch.randn_like(h) for h in yh]
            ((yl * ws[0]).sum() + sum((h * w).sum() for h, w in zip(yh, ws[1:]))).backward()
            prev = _ll.FUSED_LEVELS
            _ll.FUSED_LEVELS = False
            try:
                xb = x.clone().requires_grad_(True)
                yl2, yh2 = xfm(xb)
                assert 'WlDwt1dFused' not in pw.last_kernel()
                ((yl2 * ws[0]).sum() + sum((h * w).sum() for h, w in zip(yh2, ws[1:]))).backward()
            finally:
                _ll.FUSED_LEVELS = prev
            assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())
            assert float((yl - yl2).detach().abs().max()) <= 1e-5 * float(yl2.detach().abs().max())
