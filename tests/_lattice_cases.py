"""The lattice variant of the analysis strip kernel (csrc/wl_lattice.h, WlAfbStrip<.., QMF = 1, LAT = 1>): the column pass as
K = L/2 plane rotations whose coefficients a one-thread kernel derives on the device from the column bank as it is at call
time, accepted only if they reproduce that bank.  Every case against the ORACLE on the taps the module holds.
Shared by the emulator tests (device 'cpu' under emu_backend.emulated()) and the -m gpu tests."""
import numpy as np
import torch

import pytorch_wavelets_amd as pw
from oracle import wavelet_oracle as wo
from pytorch_wavelets_amd import filters as F


def _flat(b):
    return b.detach().cpu().double().numpy().ravel()


def _rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _is_lattice(name):
    if 'WlAfbStrip<' not in name:
        return False
    args = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]   # <T, L, QMF = 0, LAT = 0>
    return len(args) >= 4 and args[3] == '1'


def _run(xfm, x, mode, tol, what, want_lattice=True):
    c0 = pw.launch_count()
    yl, yh = xfm(x)
    ks = pw.kernels_since(c0)
    if want_lattice:   # the examination of the banks, the lattice kernel, its armed two-bank fallback
        assert len(ks) == 3 and ks[0].startswith('WlTapPrep') and _is_lattice(ks[1]) and ks[2].endswith('(armed fallback)'), (what, ks)
    oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), 1, _flat(xfm.h0_col), _flat(xfm.h1_col),
                              _flat(xfm.h0_row), _flat(xfm.h1_row), mode)
    e = max(_rel(yl, oyl), _rel(yh[0], oyh[0]))
    assert e <= tol, (what, e)
    return e


LATTICE_WAVES = [('db6', 'symmetric'), ('db7', 'zero'), ('db8', 'periodization'), ('sym8', 'reflect'), ('db10', 'periodic'),
                 ('coif3', 'symmetric'), ('coif2', 'periodization'), ('sym6', 'zero'), ('db9', 'symmetric')]


def check_lattice_vs_oracle(dev, wave, mode, shape=(2, 2, 72, 288), dtype=torch.float32):
    """One analysis level of a long orthogonal filter on the (forced) strip kernel: the lattice variant, float32: 1e-5 of the
    largest coefficient (measured 2-3e-7: as accurate as the direct sum)."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(29)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        x = torch.tensor(rng.randn(*shape), dtype=dtype, device=dev)
        xfm = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev).to(dtype)
        # (18 taps: no lattice instantiation - nine delay slots would need nine unrolled half-batches - the QMF variant runs)
        return _run(xfm, x, mode, 1e-5 if dtype == torch.float32 else 3e-3, (wave, mode), want_lattice=len(F.dwt_analysis_taps(wave)[0]) != 18)
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def check_lattice_rejections(dev, dtype=torch.float32, shape=(1, 2, 64, 288), tol=1e-5):
    """Banks the host's quadrature-mirror hint passes but that are no orthogonal pair (or not the pair the device finds when the
    kernel runs): the one-thread examination rejects them and the armed two-bank variant does the work - equal to the oracle
    on the taps in the buffers."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(31)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        x = torch.tensor(rng.randn(*shape), dtype=dtype, device=dev)
        L = 16
        sign = np.array([1.0, -1.0] * (L // 2))
        # 1. a random lowpass with its exact mirror as highpass: QMF holds (the hint is given), the pair is not orthogonal
        lo = rng.randn(L)
        hi = sign * lo[::-1]
        # (the constructor reverses what it is given: hand it the reversed pair so that the STORED pair is (lo, hi))
        xfm = pw.DWTForward(J=1, wave=(lo[::-1].copy(), hi[::-1].copy()), mode='symmetric').to(dev).to(dtype)
        assert ops.is_qmf_pair(xfm.h0_col, xfm.h1_col), 'the test wants a mirror pair in the buffers'
        _run(xfm, x, 'symmetric', tol, 'random mirror pair')
        # 2. db8 with its lowpass (and, mirrored, its highpass) perturbed by 1e-3: still a mirror pair, orthogonal to 1e-3 only
        h0, h1 = (np.asarray(v, dtype=np.float64) for v in F.dwt_analysis_taps('db8'))
        dec_lo = h0[::-1].copy()
        dec_lo[3] += 1e-3
        stored_lo = dec_lo[::-1]
        stored_hi = sign * stored_lo[::-1]
        xfm = pw.DWTForward(J=1, wave=(dec_lo, stored_hi[::-1].copy()), mode='periodization').to(dev).to(dtype)
        assert ops.is_qmf_pair(xfm.h0_col, xfm.h1_col)
        _run(xfm, x, 'periodization', tol, 'db8 perturbed by 1e-3')
        # 3. the column bank edited through `.data` AFTER a lattice launch (invisible to the host's cache key)
        xfm = pw.DWTForward(J=1, wave='db8', mode='symmetric').to(dev).to(dtype)
        _run(xfm, x, 'symmetric', tol, 'pristine db8')
        xfm.h0_col.data[0, 0, 5, 0] += 0.25
        _run(xfm, x, 'symmetric', tol, 'h0_col.data[...] +=')
        xfm = pw.DWTForward(J=1, wave='db8', mode='symmetric').to(dev).to(dtype)
        _run(xfm, x, 'symmetric', tol, 'pristine db8 (2)')
        xfm.h1_row.data.mul_(-1.0)
        _run(xfm, x, 'symmetric', tol, 'h1_row.data.mul_(-1)')
        # 4. NaN taps: rejected, and the two-bank variant propagates them like the reference would
        xfm = pw.DWTForward(J=1, wave='db8', mode='symmetric').to(dev).to(dtype)
        xfm.h0_col.data[0, 0, 2, 0] = float('nan')
        yl, yh = xfm(x)
        assert bool(torch.isnan(yl).any())
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def check_lattice_float16_module(dev, shape=(1, 2, 64, 2048)):
    """BASELINE configs[4]'s geometry at one level: a `.half()` module holds float16-ROUNDED taps, no exact orthogonal pair any
    more.  The lattice is accepted to a quarter unit in the last place of the storage type and computes the orthogonal bank
    nearest to the rounded taps: against the oracle on the rounded taps AND against the oracle on the float32 table (the
    reference's expected value, oracle/pin_fp16_config5.py) within the float16 tolerance of the suite."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(37)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        x = torch.tensor(rng.randn(*shape), device=dev).half()
        xfm = pw.DWTForward(J=1, wave='db8', mode='periodization').to(dev).half()
        _run(xfm, x, 'periodization', 2e-3, 'float16 module, rounded taps')
        yl, yh = xfm(x)
        h0, h1 = F.dwt_analysis_taps('db8')
        oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), 1, h0, h1, h0, h1, 'periodization')
        assert _rel(yl, oyl) <= 2e-3 and _rel(yh[0], oyh[0]) <= 2e-3, (_rel(yl, oyl), _rel(yh[0], oyh[0]))
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def _is_lattice_syn(name):
    if 'WlSfbStrip<' not in name:
        return False
    args = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]   # <T, L, SODD, QMF = 0, LAT = 0>
    return len(args) >= 5 and args[4] == '1'


def _run_inv(ifm, yl, yh, mode, tol, what, want_lattice=True):
    c0 = pw.launch_count()
    r = ifm((yl, yh))
    ks = pw.kernels_since(c0)
    if want_lattice:
        assert len(ks) == 3 and ks[0].startswith('WlTapPrep') and _is_lattice_syn(ks[1]) and ks[2].endswith('(armed fallback)'), (what, ks)
    want = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh],
                          _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), mode)
    e = _rel(r, want)
    assert e <= tol, (what, e)
    return e


def check_lattice_inverse_vs_oracle(dev, wave, mode, shape=(2, 2, 72, 288), dtype=torch.float32):
    """One synthesis level of a long orthogonal filter on the (forced) strip kernel: the lattice variant (the transposed
    recurrence), against the oracle on the module's taps."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(41)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        h0, h1 = F.dwt_analysis_taps(wave)
        oyl, oyh = wo.dwt_forward(rng.randn(*shape), 1, h0, h1, h0, h1, mode)
        yl = torch.tensor(oyl, dtype=dtype, device=dev)
        yh = [torch.tensor(v, dtype=dtype, device=dev) for v in oyh]
        ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dtype)
        return _run_inv(ifm, yl, yh, mode, 1e-5 if dtype == torch.float32 else 3e-3, (wave, mode), want_lattice=len(h0) != 18)
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def check_lattice_inverse_rejections(dev, dtype=torch.float32, shape=(1, 2, 64, 288), tol=1e-5):
    """Synthesis banks that pass the host's mirror-pair hint but are no orthogonal pair, or were edited through `.data`: the
    device rejects the lattice, the armed two-bank variant does the work - equal to the oracle on the taps in the buffers."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(43)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        h0, h1 = F.dwt_analysis_taps('db8')
        oyl, oyh = wo.dwt_forward(rng.randn(*shape), 1, h0, h1, h0, h1, 'symmetric')
        yl = torch.tensor(oyl, dtype=dtype, device=dev)
        yh = [torch.tensor(v, dtype=dtype, device=dev) for v in oyh]
        L = 16
        sign = np.array([1.0, -1.0] * (L // 2))
        lo = rng.randn(L)
        ifm = pw.DWTInverse(wave=(lo, sign * lo[::-1]), mode='symmetric').to(dev).to(dtype)
        assert ops.is_qmf_pair(ifm.g0_col, ifm.g1_col), 'the test wants a mirror pair in the buffers'
        _run_inv(ifm, yl, yh, 'symmetric', tol, 'random mirror pair')
        ifm = pw.DWTInverse(wave='db8', mode='symmetric').to(dev).to(dtype)
        _run_inv(ifm, yl, yh, 'symmetric', tol, 'pristine db8')
        ifm.g0_col.data[0, 0, 4, 0] -= 0.125
        _run_inv(ifm, yl, yh, 'symmetric', tol, 'g0_col.data[...] -=')
        ifm = pw.DWTInverse(wave='db8', mode='symmetric').to(dev).to(dtype)
        _run_inv(ifm, yl, yh, 'symmetric', tol, 'pristine db8 (2)')
        ifm.g1_row.data.mul_(0.5)
        _run_inv(ifm, yl, yh, 'symmetric', tol, 'g1_row.data.mul_')
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def check_lattice_levels_share_one_examination(dev, shape=(1, 2, 128, 1024)):
    """The levels of ONE module call share their float32 taps and the one-thread examination of the banks (policy bit 2 of the
    strip entry points): J = 2 forward and inverse of a `.half()` module = one WlTapPrep launch each, two lattice launches, two
    armed fallbacks - and the result against the oracle on the module's taps."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(47)
    prev = ops.STREAM_FORCE, _ll.FUSED_LEVELS
    ops.STREAM_FORCE, _ll.FUSED_LEVELS = True, False
    try:
        x = torch.tensor(rng.randn(*shape), device=dev).half()
        xfm = pw.DWTForward(J=2, wave='db8', mode='periodization').to(dev).half()
        ifm = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half()
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        ks = pw.kernels_since(c0)
        assert [k.startswith('WlTapPrep') for k in ks] == [True, False, False, False, False] and _is_lattice(ks[1]) and _is_lattice(ks[3]), ks
        oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), 2, _flat(xfm.h0_col), _flat(xfm.h1_col),
                                  _flat(xfm.h0_row), _flat(xfm.h1_row), 'periodization')
        assert _rel(yl, oyl) <= 3e-3 and all(_rel(a, b) <= 3e-3 for a, b in zip(yh, oyh))
        c0 = pw.launch_count()
        r = ifm((yl, yh))
        ks = pw.kernels_since(c0)
        assert [k.startswith('WlTapPrep') for k in ks] == [True, False, False, False, False] and _is_lattice_syn(ks[1]) and _is_lattice_syn(ks[3]), ks
        want = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh],
                              _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), 'periodization')
        assert _rel(r, want) <= 3e-3
    finally:
        ops.STREAM_FORCE, _ll.FUSED_LEVELS = prev


def check_tile_kernels_14_18_taps(dev, wave, shape=(2, 2, 100, 104)):
    """db7 / sym7 (14 taps) and db9 / sym9 (18 taps) on the compile-time-tap tile kernels (round 5: before, every narrow level
    of theirs ran on the run-time-tap kernel): J = 2 forward and inverse, every mode, float32 + float16, against the oracle."""
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(53)
    h0, h1 = F.dwt_analysis_taps(wave)
    g0, g1 = F.dwt_synthesis_taps(wave)
    L = len(h0)
    prev = _ll.FUSED_LEVELS
    _ll.FUSED_LEVELS = False      # (one launch per level: with enough planes 14 taps now have a fused lattice kernel of their own)
    try:
        _tile_14_18_modes(dev, wave, shape, rng, h0, h1, g0, g1, L)
    finally:
        _ll.FUSED_LEVELS = prev


def _tile_14_18_modes(dev, wave, shape, rng, h0, h1, g0, g1, L):
    for mode in ('zero', 'symmetric', 'periodization', 'reflect', 'periodic'):
        for dt, tol in ((torch.float32, 1e-5), (torch.float16, 4e-3)):
            x = torch.tensor(rng.randn(*shape)).to(dt).to(dev)
            oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), 2, h0, h1, h0, h1, mode)
            xfm = pw.DWTForward(J=2, wave=wave, mode=mode).to(dev)
            ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev)
            c0 = pw.launch_count()
            yl, yh = xfm(x)
            ks = pw.kernels_since(c0)
            assert any(k.startswith('WlAfbTile<') and ', %d' % L in k for k in ks), (wave, mode, ks)
            c0 = pw.launch_count()
            r = ifm((yl, yh))
            ks = pw.kernels_since(c0)
            assert any(k.startswith('WlSfbTile<') and ', %d,' % L in k for k in ks), (wave, mode, ks)
            orec = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh], g0, g1, g0, g1, mode)
            errs = [_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)] + [_rel(r, orec)]
            assert max(errs) <= tol, (wave, mode, dt, errs)


ROWS_LATTICE_CASES = [('db5', 'symmetric', 3), ('db6', 'zero', 3), ('sym7', 'reflect', 2), ('db8', 'symmetric', 3), ('sym8', 'zero', 2),
                      ('db10', 'symmetric', 2), ('coif2', 'reflect', 3), ('db7', 'symmetric', 3), ('db9', 'symmetric', 2), ('coif3', 'zero', 3)]


def _is_lattice_rows(name):
    if 'WlAfbRows<' not in name or not name.rstrip().endswith('>'):
        return False
    args = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]   # <T, L, PPR, D, SAME, LAT[, ODD]>
    return len(args) >= 6 and args[4] == '1' and args[5] == '1'


def check_rows_lattice_vs_oracle(dev, wave, mode, J, shape=(2, 3, 128, 256), dtype=torch.float32):
    """The fused multi-level analysis kernel in its lattice form (WlAfbRows<.., SAME = 1, LAT = 1>, 10-20 taps: the only fused
    form of 14, 16 and 20 taps): DWTForward on the (forced) streaming kernel against the oracle on the module's taps."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(59)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = 1, 0      # (the engine's size policy for the lattice at <= 12 taps: off)
    try:
        x = torch.tensor(rng.randn(*shape), device=dev).to(dtype)
        xfm = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev).to(dtype)
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        ks = pw.kernels_since(c0)
        prim = [k for k in ks if not k.endswith(')')]
        # (the finest levels in one lattice launch; a coarsest level whose rows are no whole 16-byte pieces follows on its own kernel)
        assert prim and _is_lattice_rows(prim[0]) and ks[0].startswith('WlTapPrep') and any(k.endswith('(armed fallback)') for k in ks), ks
        oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), J, _flat(xfm.h0_col), _flat(xfm.h1_col),
                                  _flat(xfm.h0_row), _flat(xfm.h1_row), mode)
        tol = 1e-5 if dtype == torch.float32 else 4e-3
        errs = [_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)]
        assert max(errs) <= tol, (wave, mode, J, errs)
        return max(errs)
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


def check_rows_lattice_rejections(dev, dtype=torch.float32, shape=(2, 2, 96, 256), tol=1e-5):
    """Banks the fused lattice launch cannot take - the device rejects them and the armed two-bank kernel does the work (for 16
    taps that is the direct-form kernel that exists for this purpose only): a random mirror pair on both axes, the column bank
    edited through `.data` (the hints go stale), the banks of the two axes differing through `.data`."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(61)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = 1, 0      # (the engine's size policy for the lattice at <= 12 taps: off)
    try:
        x = torch.tensor(rng.randn(*shape), device=dev).to(dtype)

        def run(xfm, J, what):
            c0 = pw.launch_count()
            yl, yh = xfm(x)
            ks = pw.kernels_since(c0)
            assert any(_is_lattice_rows(k) for k in ks) and any(k.endswith('(armed fallback)') for k in ks), (what, ks)
            oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), J, _flat(xfm.h0_col), _flat(xfm.h1_col),
                                      _flat(xfm.h0_row), _flat(xfm.h1_row), 'symmetric')
            errs = [_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)]
            assert max(errs) <= tol, (what, errs)

        for L in (12, 16):
            sign = np.array([1.0, -1.0] * (L // 2))
            lo = rng.randn(L) / 3
            hi = sign * lo[::-1]
            xfm = pw.DWTForward(J=2, wave=(lo[::-1].copy(), hi[::-1].copy()), mode='symmetric').to(dev).to(dtype)
            assert ops.is_qmf_pair(xfm.h0_col, xfm.h1_col)
            run(xfm, 2, 'random mirror pair, %d taps' % L)
        for wave in ('db6', 'db8'):
            xfm = pw.DWTForward(J=2, wave=wave, mode='symmetric').to(dev).to(dtype)
            run(xfm, 2, 'pristine ' + wave)
            xfm.h0_col.data[0, 0, 3, 0] += 0.125
            run(xfm, 2, wave + ': h0_col.data[...] +=')
            xfm = pw.DWTForward(J=2, wave=wave, mode='symmetric').to(dev).to(dtype)
            run(xfm, 2, 'pristine (2) ' + wave)
            xfm.h1_row.data.mul_(2.0)
            run(xfm, 2, wave + ': h1_row.data.mul_')
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


def _is_lattice_irows(name):
    if 'WlSfbRows<' not in name:
        return False
    args = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]   # <T, L, LAT = 0>
    return len(args) >= 3 and args[2] == '1'


def check_irows_lattice_vs_oracle(dev, wave, mode, J, shape=(2, 3, 128, 256), dtype=torch.float32, require=True):
    """The fused multi-level synthesis kernel in its lattice form (WlSfbRows<T, L, LAT = 1>, 10-20 taps: the only fused form of
    14, 16 and 20 taps): DWTInverse on the (forced) streaming kernel against the oracle on the module's taps."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(67)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = 1, 0      # (the engine's size policy for the lattice at <= 12 taps: off)
    try:
        h0, h1 = F.dwt_analysis_taps(wave)
        oyl, oyh = wo.dwt_forward(rng.randn(*shape), J, h0, h1, h0, h1, mode)
        yl = torch.tensor(oyl, device=dev).to(dtype)
        yh = [torch.tensor(v, device=dev).to(dtype) for v in oyh]
        ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dtype)
        c0 = pw.launch_count()
        r = ifm((yl, yh))
        ks = pw.kernels_since(c0)
        # (require = False: float16 levels whose coefficient rows are no whole 4-byte words - 263 halves - stay off the fused kernel)
        if require or any(_is_lattice_irows(k) for k in ks):
            assert any(_is_lattice_irows(k) for k in ks) and any(k.startswith('WlTapPrep') for k in ks) and any(k.endswith('(armed fallback)') for k in ks), ks
        want = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh],
                              _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), mode)
        e = _rel(r, want)
        assert e <= (1e-5 if dtype == torch.float32 else 4e-3), (wave, mode, J, e)
        return e
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


def check_irows_lattice_rejections(dev, dtype=torch.float32, shape=(2, 2, 96, 256), tol=1e-5):
    """Synthesis banks the fused lattice launch cannot take: a random mirror pair, banks edited through `.data` - the device
    rejects them, the armed four-bank kernel does the work; equal to the oracle on the taps in the buffers."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(71)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = 1, 0      # (the engine's size policy for the lattice at <= 12 taps: off)
    try:
        def run(ifm, yl, yh, what):
            c0 = pw.launch_count()
            r = ifm((yl, yh))
            ks = pw.kernels_since(c0)
            assert any(_is_lattice_irows(k) for k in ks) and any(k.endswith('(armed fallback)') for k in ks), (what, ks)
            want = wo.dwt_inverse(yl.detach().cpu().double().numpy(), [h.detach().cpu().double().numpy() for h in yh],
                                  _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), 'symmetric')
            assert _rel(r, want) <= tol, (what, _rel(r, want))

        for wave in ('db6', 'db8'):
            h0, h1 = F.dwt_analysis_taps(wave)
            L = len(h0)
            oyl, oyh = wo.dwt_forward(rng.randn(*shape), 2, h0, h1, h0, h1, 'symmetric')
            yl = torch.tensor(oyl, device=dev).to(dtype)
            yh = [torch.tensor(v, device=dev).to(dtype) for v in oyh]
            sign = np.array([1.0, -1.0] * (L // 2))
            lo = rng.randn(L) / 3
            ifm = pw.DWTInverse(wave=(lo, sign * lo[::-1]), mode='symmetric').to(dev).to(dtype)
            assert ops.is_qmf_pair(ifm.g0_col, ifm.g1_col)
            run(ifm, yl, yh, 'random mirror pair, %d taps' % L)
            ifm = pw.DWTInverse(wave=wave, mode='symmetric').to(dev).to(dtype)
            run(ifm, yl, yh, 'pristine ' + wave)
            ifm.g0_row.data[0, 0, 0, 2] += 0.25
            run(ifm, yl, yh, wave + ': g0_row.data[...] +=')
            ifm = pw.DWTInverse(wave=wave, mode='symmetric').to(dev).to(dtype)
            run(ifm, yl, yh, 'pristine (2) ' + wave)
            ifm.g1_col.data.mul_(-1.0)
            run(ifm, yl, yh, wave + ': g1_col.data.mul_')
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev


# ---- a scratch block is trusted only after the LIBRARY examined it (tap_state of the *_ex entry points) -------------------------
def _poisoned_scratch(device):
    """What a recycled allocator block may hold: the OK verdict and lattice of ANOTHER bank (here: nonsense coefficients)."""
    import struct
    ok = struct.unpack('f', struct.pack('I', 0x4c415431))[0]       # WL_LAT_OK of csrc/wl_lattice.h
    return torch.tensor([ok, 3.0] + [0.7 * (-1) ** k * (k + 1) for k in range(14)], dtype=torch.float32, device=device)


def check_unexamined_scratch_is_never_trusted(dev, shape=(1, 2, 64, 1024)):
    """a strip level that makes NO use of the device scratch (8 / 10 taps) must not make the fused lattice
    launch of the next levels skip its examination.  DWTForward(J = 3) / DWTInverse of 8- and 10-tap wavelets on a 1024-wide plane
    (level 1 on the strip kernel, the rest fused in lattice form) with every scratch block handed out POISONED (the verdict word
    of an accepted factorisation + nonsense coefficients): WlTapPrep must run in front of the lattice kernel, and the result is the
    oracle's."""
    from pytorch_wavelets_amd import ops
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    rng = np.random.RandomState(61)
    prev = ops.STREAM_FORCE, ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops._new_tap_scratch, _ll.WIDE_ONE_LEVEL
    ops.STREAM_FORCE, ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML = True, 1, 0, 0
    ops._new_tap_scratch = _poisoned_scratch
    _ll.WIDE_ONE_LEVEL = 256
    try:
        for wave in ('db5', 'db4', 'db6'):
            x = torch.tensor(rng.randn(*shape), dtype=torch.float32, device=dev)
            xfm = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev)
            ifm = pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
            c0 = pw.launch_count()
            yl, yh = xfm(x)
            ks = pw.kernels_since(c0)
            lat = [i for i, k in enumerate(ks) if _is_lattice_rows(k) or _is_lattice(k)]
            for i in lat:    # every lattice launch of the call sits behind an examination made in THIS call
                assert any(k.startswith('WlTapPrep') for k in ks[:i]), (wave, ks)
            oyl, oyh = wo.dwt_forward(x.cpu().double().numpy(), 3, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row), _flat(xfm.h1_row), 'symmetric')
            e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
            assert e <= 1e-5, (wave, 'forward', e, ks)
            c0 = pw.launch_count()
            r = ifm((yl, yh))
            ks = pw.kernels_since(c0)
            lat = [i for i, k in enumerate(ks) if _is_lattice_irows(k) or _is_lattice_syn(k)]
            for i in lat:
                assert any(k.startswith('WlTapPrep') for k in ks[:i]), (wave, ks)
            want = wo.dwt_inverse(oyl, oyh, _flat(ifm.g0_col), _flat(ifm.g1_col), _flat(ifm.g0_row), _flat(ifm.g1_row), 'symmetric')
            e = _rel(r, want)
            assert e <= 1e-5, (wave, 'inverse', e, ks)
    finally:
        ops.STREAM_FORCE, ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops._new_tap_scratch, _ll.WIDE_ONE_LEVEL = prev


def check_tap_state_contract(dev):
    """The C ABI directly: wl_dwt2d_analysis_stream_ex with a poisoned scratch.  10 taps (no use for the scratch): *tap_state stays 0.
    12 taps with the hint: the library examines (bit 0 set), a second call with that state skips the examination, and a state the
    CALLER forged (bit 0 on a poisoned block) is the caller's contract to keep - the library documents it, the Python layer never does it."""
    import ctypes
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(5)
    x = torch.tensor(rng.randn(1, 2, 32, 288), dtype=torch.float32, device=dev)
    for wave, uses in (('db5', False), ('db6', True)):
        h0, h1 = (np.asarray(v, dtype=np.float64) for v in F.dwt_analysis_taps(wave))
        L = len(h0)
        taps = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]    # (stored order: what the modules' buffers hold)
        scratch = _poisoned_scratch(dev)
        st = ctypes.c_int(0)
        Kh, Kw = ops.coeff_len(32, L, 1), ops.coeff_len(288, L, 1)
        ll = torch.empty(1, 2, Kh, Kw, dtype=torch.float32, device=dev)
        hs = torch.empty(1, 2, 3, Kh, Kw, dtype=torch.float32, device=dev)

        def call():
            c0 = pw.launch_count()
            rc = ops._call('wl_dwt2d_analysis_stream_ex', x, x.data_ptr(), 32 * 288, 288, ll.data_ptr(), Kh * Kw, Kw, hs.data_ptr(), 0, 2, 32, 288,
                           taps[0].data_ptr(), taps[1].data_ptr(), taps[2].data_ptr(), taps[3].data_ptr(), L, 1, 1 | 2, scratch.data_ptr(),
                           ctypes.byref(st), ops._stream(x))
            assert rc == 0, rc
            return pw.kernels_since(c0)
        ks = call()
        assert st.value == (1 if uses else 0), (wave, st.value, ks)
        assert any(k.startswith('WlTapPrep') for k in ks) == uses, (wave, ks)
        oyl, oyh = wo.dwt_forward(x.cpu().double().numpy(), 1, taps[0].cpu().double().numpy(), taps[1].cpu().double().numpy(),
                                  taps[2].cpu().double().numpy(), taps[3].cpu().double().numpy(), 'symmetric')
        assert max(_rel(ll, oyl), _rel(hs, oyh[0])) <= 1e-5, (wave, _rel(ll, oyl), _rel(hs, oyh[0]), ks)
        ks = call()
        assert not any(k.startswith('WlTapPrep') for k in ks), (wave, ks)          # examined once (or never needed)
        assert max(_rel(ll, oyl), _rel(hs, oyh[0])) <= 1e-5, wave


# ---- round 6: LL rings sized exactly (WlAfbRows<.., NP2 = 1>): three levels of a long filter on 512 columns in symmetric / reflect mode ---------
NP2_CASES = [('db8', 'symmetric'), ('db7', 'reflect'), ('db6', 'symmetric'), ('sym8', 'reflect'), ('db10', 'zero')]


def check_rows_exact_rings(dev, wave, mode, shape=(1, 2, 200, 512), dtype=torch.float32, planes_cut=False, require_np2=True):
    """DWTForward J = 3 of a 12- to 20-tap orthogonal wavelet on 512-column planes: with power-of-two LL rings the three levels do not fit the
    80 KiB of two workgroups per CU and the launcher takes the instantiation whose rings have exactly the rows the simulated schedule needs
    (slot = row mod rows).  ONE launch does the work; against the oracle."""
    from pytorch_wavelets_amd import ops
    rng = np.random.RandomState(101)
    prev = ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS
    ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = (2 if planes_cut else 1), 0
    try:
        x = torch.tensor(rng.randn(*shape), device=dev).to(dtype)
        xfm = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev).to(dtype)
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        prim = [k for k in pw.kernels_since(c0) if not k.endswith(')')]
        args = [a.strip() for a in prim[0][prim[0].index('<') + 1:prim[0].rindex('>')].split(',')] if prim else []
        assert len(prim) == 1 and prim[0].startswith('WlAfbRows<') and (not require_np2 or (len(args) == 8 and args[7] == '1')), prim   # <T, L, PPR, D, SAME, LAT, ODD, NP2>
        oyl, oyh = wo.dwt_forward(x.detach().cpu().double().numpy(), 3, _flat(xfm.h0_col), _flat(xfm.h1_col), _flat(xfm.h0_row), _flat(xfm.h1_row), mode)
        e = max([_rel(yl, oyl)] + [_rel(a, b) for a, b in zip(yh, oyh)])
        assert e <= (1e-5 if dtype == torch.float32 else 4e-3), (wave, mode, e)
        return e
    finally:
        ops.FUSED_STRIPS, ops.LATTICE_MIN_ELEMS = prev
