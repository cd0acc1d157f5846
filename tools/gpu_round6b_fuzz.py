"""Random pyramids of LONG orthogonal filters (12-20 taps) on 400-760-column planes in zero / symmetric / reflect mode on the real GPU: the fused
analysis kernel (forced; whole and cut planes) - whose LL rings are sized exactly where the power-of-two layout does not fit (WlAfbRows<.., NP2 = 1>,
round 6) - against the per-level kernels (dwt.lowlevel.FUSED_LEVELS = False), and every fifth case against the ORACLE on two planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel as ll
from oracle import wavelet_oracle as wo
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
waves = ['db6', 'db7', 'db8', 'db9', 'db10', 'sym6', 'sym7', 'sym8', 'coif2', 'coif3']
bad = np2 = fused3 = orc = 0
flat = lambda t: t.detach().cpu().double().numpy().ravel()
def err(p, q):
    return float((p.float() - q.float()).abs().max() / max(1e-6, float(q.float().abs().max())))
for seed in range(n):
    rng = np.random.RandomState(14000 + seed)
    wave = waves[rng.randint(len(waves))]
    mode = ('zero', 'symmetric', 'reflect')[rng.randint(3)]
    dt = torch.float16 if rng.rand() < 0.25 else torch.float32
    tol = 4e-3 if dt == torch.float16 else 1e-5
    q = 8 if dt == torch.float16 else 4
    H = int(rng.randint(150, 420)); W = q * int(rng.randint(400 // q, 760 // q))
    planes = (int(rng.randint(1, 3)), int(rng.randint(1, 5)))
    x = torch.randn(planes[0], planes[1], H, W, device=dev).to(dt)
    xfm = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev).to(dt)
    res = {}
    for fused in (False, True):
        ll.FUSED_LEVELS = fused
        ops.FUSED_STRIPS = int(rng.randint(1, 3)) if fused else 0
        ops.LATTICE_MIN_ELEMS = 0 if fused else 40000000
        ops._FUSED_DECLINED.clear()
        with torch.no_grad():
            c0 = pw.launch_count(); yl, yh = xfm(x); ks = [k for k in pw.kernels_since(c0) if not k.endswith(')')]
        res[fused] = (yl, yh, ks)
    ll.FUSED_LEVELS = True; ops.FUSED_STRIPS = 0; ops.LATTICE_MIN_ELEMS = 40000000
    (yl0, yh0, _), (yl1, yh1, ks) = res[False], res[True]
    fused3 += len(ks) == 1 and ks[0].startswith('WlAfbRows')
    np2 += any(k.startswith('WlAfbRows') and k.rstrip('>').endswith(', 1') and k.count(',') == 7 for k in ks)
    es = [err(yl1, yl0)] + [err(a, b) for a, b in zip(yh1, yh0)]
    if seed % 5 == 0:
        orc += 1
        oyl, oyh = wo.dwt_forward(x[:1, :2].cpu().double().numpy(), 3, flat(xfm.h0_col), flat(xfm.h1_col), flat(xfm.h0_row), flat(xfm.h1_row), mode)
        es += [err(yl1[:1, :2].cpu(), torch.tensor(oyl))] + [err(a[:1, :2].cpu(), torch.tensor(b)) for a, b in zip(yh1, oyh)]
    if not max(es) < tol:
        bad += 1
        print('BAD', seed, wave, mode, dt, planes, H, W, ['%.1e' % e for e in es], ks)
print('round-6 long-filter fuzz: %d cases (%d as ONE fused launch, %d of them with exactly sized rings, %d also against the oracle), %d mismatches' % (n, fused3, np2, orc, bad))
