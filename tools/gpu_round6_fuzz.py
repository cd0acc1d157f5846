"""Random shapes on the real GPU for round 6's fused PERIODIZATION paths (several levels per launch: csrc/wl_dwt_rows.h ODD / negative feeds,
csrc/wl_idwt_rows.h PER) through the modules: DWTForward / DWTInverse with the fused kernels forced (whole planes, cut planes, the policy)
against the level-by-level ladder of round 5 in the same process (ops.ROWS_PER / IROWS_PER off) - and every fifth case against the ORACLE on
two planes.  Also the gradient of the forward (an inverse transform with the analysis taps).  Prints the failures (none expected) and a summary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from oracle import wavelet_oracle as wo
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
waves = ['haar', 'db2', 'db3', 'db4', 'db5', 'db6', 'db7', 'db8', 'db9', 'db10', 'sym4', 'coif1', 'coif2', 'bior2.2', 'bior4.4', 'bior1.3']
bad = fused_f = fused_i = orc = 0
flat = lambda t: t.detach().cpu().double().numpy().ravel()
def err(p, q):
    return float((p.float() - q.float()).abs().max() / max(1e-6, float(q.float().abs().max())))
for seed in range(n):
    rng = np.random.RandomState(12000 + seed)
    wave = waves[rng.randint(len(waves))]
    J = int(rng.randint(1, 4))
    dt = torch.float16 if rng.rand() < 0.3 else torch.float32
    tol = 4e-3 if dt == torch.float16 else 1e-5
    q = 8 if dt == torch.float16 else 4
    mult = max(q, 1 << J)
    H = (1 << J) * int(rng.randint(3, 80)); W = mult * int(rng.randint(2, 768 // mult))
    planes = (int(rng.randint(1, 4)), int(rng.randint(1, 6)))
    strips = int(rng.randint(3))
    x = torch.randn(planes[0], planes[1], H, W, device=dev).to(dt)
    xfm = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev).to(dt)
    ifm = pw.DWTInverse(wave=wave, mode='periodization').to(dev).to(dt)
    res = {}
    for fused in (False, True):
        ops.ROWS_PER = ops.IROWS_PER = fused
        ops.FUSED_STRIPS = strips if fused else 0
        ops.LATTICE_MIN_ELEMS, ops.LATTICE_MIN_ELEMS_ML, ops.IROWS_F16_MAXL = (0, 0, 99) if fused else (40000000, 16000000, 8)
        ops._FUSED_DECLINED.clear()
        xa = x.clone().requires_grad_(dt == torch.float32)
        c0 = pw.launch_count(); yl, yh = xfm(xa); kf = pw.kernels_since(c0)
        c0 = pw.launch_count(); r = ifm((yl.detach(), [h.detach() for h in yh])); ki = pw.kernels_since(c0)
        g = None
        if dt == torch.float32:
            wl_, wh_ = torch.ones_like(yl), [torch.full_like(h, 0.5) for h in yh]
            g, = torch.autograd.grad((yl * wl_).sum() + sum((h * w).sum() for h, w in zip(yh, wh_)), xa)
        res[fused] = (yl.detach(), [h.detach() for h in yh], r, g, kf, ki)
    (yl0, yh0, r0, g0, _, _), (yl1, yh1, r1, g1, kf, ki) = res[False], res[True]
    fused_f += any('WlAfbRows' in k for k in kf); fused_i += any('WlSfbRows' in k for k in ki)
    es = [err(yl1, yl0)] + [err(a, b) for a, b in zip(yh1, yh0)] + [err(r1, r0)] + ([err(g1, g0)] if g0 is not None else [])
    if seed % 5 == 0:
        orc += 1
        xs = x[:1, :2].cpu().double().numpy()
        oyl, oyh = wo.dwt_forward(xs, J, flat(xfm.h0_col), flat(xfm.h1_col), flat(xfm.h0_row), flat(xfm.h1_row), 'periodization')
        orr = wo.dwt_inverse(yl1[:1, :2].cpu().double().numpy(), [h[:1, :2].cpu().double().numpy() for h in yh1], flat(ifm.g0_col), flat(ifm.g1_col),
                             flat(ifm.g0_row), flat(ifm.g1_row), 'periodization')
        es += [err(yl1[:1, :2].cpu(), torch.tensor(oyl)), err(r1[:1, :2].cpu(), torch.tensor(orr))] + [err(a[:1, :2].cpu(), torch.tensor(b)) for a, b in zip(yh1, oyh)]
    if not max(es) < tol:
        bad += 1
        print('BAD', seed, wave, J, dt, planes, H, W, 'strips', strips, ['%.1e' % e for e in es], kf, ki)
print('round-6 periodization fuzz: %d cases (%d forward / %d inverse through the fused kernels, %d also against the oracle), %d mismatches' % (n, fused_f, fused_i, orc, bad))
