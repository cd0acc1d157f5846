"""Random pyramids on the real GPU for what round 5 added late: several planes per workgroup on narrow strip levels, short row
segments, the fused analysis on row-padded / 3 KiB rows, the ladder that sends a wide lone synthesis level to the strip kernel.
Every case runs DWTForward / DWTInverse twice - as the engine dispatches it (half the cases with the strip kernels forced, so that
narrow levels with many planes pack) and on the per-level TILE kernels alone (no streaming, no fusion: wl_set_option no_stream +
FUSED_LEVELS off) - and compares all coefficients and the reconstruction; the gradient of a random third of the cases too.
Prints the failures (none expected), the kernels seen and a summary line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel as ll_
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
waves = ['haar', 'db2', 'db3', 'db4', 'db5', 'db6', 'db7', 'db8', 'db10', 'sym4', 'sym8', 'coif2', 'bior2.2', 'bior4.4']
modes = ['zero', 'symmetric', 'reflect', 'periodic', 'periodization']
be = ops._backend()
bad = packed = padded = 0
seen = set()
def run(xfm, ifm, x, grad):
    xx = x.clone().requires_grad_(grad)
    c0 = pw.launch_count()
    yl, yh = xfm(xx)
    rec = ifm((yl, yh))
    ks = pw.kernels_since(c0)
    g = None
    if grad:
        (yl.square().sum() + sum(h.square().sum() for h in yh)).backward()
        g = xx.grad
    return yl.detach(), [h.detach() for h in yh], rec.detach(), g, ks
for seed in range(n):
    rng = np.random.RandomState(12000 + seed)
    wave, mode = waves[rng.randint(len(waves))], modes[rng.randint(len(modes))]
    dt = torch.float16 if rng.rand() < 0.3 else torch.float32
    kind = rng.randint(4)
    if kind == 0:      # many narrow planes: the strip kernels forced, planes packed
        planes, H, W = int(rng.randint(600, 2600)), 4 * int(rng.randint(8, 40)), 4 * int(rng.randint(16, 130))
    elif kind == 1:    # rows of 2-3 KiB and around: the fused analysis with three pieces per row / a strip level + padded ll
        planes, H, W = int(rng.randint(100, 400)), int(rng.randint(40, 200)), int(rng.randint(500, 800))
    elif kind == 2:    # 1024- / 2048-wide pyramids
        planes, H, W = int(rng.randint(96, 260)), int(rng.randint(40, 160)), int(rng.choice([1024, 1028, 1100, 2048, 1536, 1000]))
    else:              # config-5-like: wide float16 / long filters, deep
        planes, H, W = int(rng.randint(64, 600)), 8 * int(rng.randint(8, 40)), 8 * int(rng.randint(32, 160))
    J = int(rng.randint(1, 5))
    if mode == 'periodization':
        H += H % 2; W += (-W) % 4
    from pytorch_wavelets_amd import filters as F_
    L = len(F_.dwt_analysis_taps(wave)[0])
    if min(H, W) >> (J - 1) < 2 * L + 2:
        J = 1
    x = torch.randn(planes // 3 + 1, 3, H, W, device=dev).to(dt)
    xfm = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev).to(dt)
    ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
    grad = rng.rand() < 0.33
    force = kind == 0 or rng.rand() < 0.3
    try:
        ops.STREAM_FORCE = force
        a = run(xfm, ifm, x, grad)
        ops.STREAM_FORCE = False
        be.wl_set_option(b'no_stream', 1); prev = ll_.FUSED_LEVELS; ll_.FUSED_LEVELS = False; sp = ops.SMALL_PLANES; ops.SMALL_PLANES = False
        ops._FUSED_DECLINED.clear()
        try:
            b = run(xfm, ifm, x, grad)
        finally:
            be.wl_set_option(b'no_stream', 0); ll_.FUSED_LEVELS = prev; ops.SMALL_PLANES = sp; ops._FUSED_DECLINED.clear()
    finally:
        ops.STREAM_FORCE = False
    for k in a[4]:
        seen.add(k.split('<')[0] + ('<..PPR3>' if 'WlAfbRows<' in k and k.split(',')[2].strip().startswith('3') else ''))
    tol = 4e-3 if dt == torch.float16 else 5e-6
    def err(p, q):
        return float((p.float() - q.float()).abs().max() / max(1e-6, float(q.float().abs().max())))
    es = [err(a[0], b[0])] + [err(p, q) for p, q in zip(a[1], b[1])] + [err(a[2], b[2])] + ([err(a[3], b[3])] if grad else [])
    if not max(es) < tol:
        bad += 1
        print('BAD', seed, wave, mode, dt, x.shape, J, 'force', force, max(es), a[4], b[4], flush=True)
print('kernels seen:', sorted(seen))
print('round-5 late fuzz: %d cases, %d mismatches' % (n, bad))
