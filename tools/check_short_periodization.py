"""Dev check (authoring container only): the engine on the host emulator and the oracle against the REAL reference for
periodization of signals shorter than the filter, odd tap counts and deep pyramids on small images.
    PYTHONPATH=tools/ref_shim:/root/reference python tools/check_short_periodization.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pytorch_wavelets as ref                      # noqa: E402
from pytorch_wavelets.dwt import lowlevel as rll    # noqa: E402
import pytorch_wavelets_amd as pw                   # noqa: E402
from pytorch_wavelets_amd.dwt import lowlevel as ell  # noqa: E402
from oracle import wavelet_oracle as wo             # noqa: E402
import emu_backend                                  # noqa: E402

torch.set_default_dtype(torch.float64)
rng = np.random.RandomState(5)
worst = 0.0


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


with emu_backend.emulated():
    # function-level afb1d / sfb1d, periodization, even and odd tap counts, lengths 1..24, both axes
    for L in (2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20):
        h0, h1 = rng.randn(L), rng.randn(L)
        t0, t1 = torch.tensor(h0[::-1].copy()), torch.tensor(h1[::-1].copy())   # tensors are taken as already reversed
        ncmp = 0
        for n in list(range(1, 13)) + [17, 24]:
            for d in (2, 3):
                shape = [2, 2, 5, 5]
                shape[d] = n
                x = torch.tensor(rng.randn(*shape))
                try:
                    r = rll.afb1d(x, t0, t1, mode='periodization', dim=d)
                except Exception as e:   # the reference itself fails (shape mismatch in its fold)
                    print('reference raises for L=%d n=%d: %s' % (L, n, str(e)[:50]))
                    continue
                ncmp += 1
                e = ell.afb1d(x, t0, t1, mode='periodization', dim=d)
                o = wo.afb1d(x.numpy(), h0[::-1], h1[::-1], 'periodization', axis=d)
                C = x.shape[1]
                rr = r.reshape(x.shape[0], C, 2, *r.shape[2:])
                err = max(rel(e.numpy(), r.numpy()), rel(o[0], rr[:, :, 0].numpy()), rel(o[1], rr[:, :, 1].numpy()))
                worst = max(worst, err)
                assert err < 1e-10, ('afb1d', L, n, d, err)
                if L % 2 == 0:
                    lo, hi = rr[:, :, 0].contiguous(), rr[:, :, 1].contiguous()
                    g0, g1 = rng.randn(L), rng.randn(L)
                    ry = rll.sfb1d(lo, hi, torch.tensor(g0), torch.tensor(g1), mode='periodization', dim=d)
                    ey = ell.sfb1d(lo, hi, torch.tensor(g0), torch.tensor(g1), mode='periodization', dim=d)
                    oy = wo.sfb1d(lo.numpy(), hi.numpy(), g0, g1, 'periodization', axis=d)
                    err = max(rel(ey.numpy(), ry.numpy()), rel(oy, ry.numpy()))
                    worst = max(worst, err)
                    assert err < 1e-10, ('sfb1d', L, n, d, err)
        assert ncmp > 20, (L, ncmp)
    print('afb1d / sfb1d periodization sweep ok, worst', worst)
    # deep pyramids on small images: levels of 1-4 samples under 12-20 taps, incl. the gradients
    for wave, J, shape in (('db6', 5, (1, 2, 50, 28)), ('db7', 5, (2, 1, 50, 28)), ('db10', 4, (1, 1, 40, 24)),
                           ('db5', 4, (1, 1, 18, 30)), ('db8', 6, (1, 1, 64, 33))):
        x = torch.tensor(rng.randn(*shape), requires_grad=True)
        rx, ri = ref.DWTForward(J=J, wave=wave, mode='periodization'), ref.DWTInverse(wave=wave, mode='periodization')
        ex, ei = pw.DWTForward(J=J, wave=wave, mode='periodization'), pw.DWTInverse(wave=wave, mode='periodization')
        ryl, ryh = rx(x)
        eyl, eyh = ex(x)
        errs = [rel(eyl.detach().numpy(), ryl.detach().numpy())] + [rel(a.detach().numpy(), b.detach().numpy()) for a, b in zip(eyh, ryh)]
        rrec, erec = ri((ryl, ryh)), ei((eyl, eyh))
        errs.append(rel(erec.detach().numpy(), rrec.detach().numpy()))
        g = torch.tensor(rng.randn(*rrec.shape))
        rdx, = torch.autograd.grad((rrec * g).sum(), x)
        edx, = torch.autograd.grad((erec * g).sum(), x)
        errs.append(rel(edx.numpy(), rdx.numpy()))
        print(wave, J, shape, 'max err %.2e' % max(errs))
        assert max(errs) < 1e-9, errs
print('ok')
