"""Random narrow planes (64 .. 260 columns, enough of them to fill the chip) through the streaming DTCWT / ScatLayer kernels (four /
two planes per workgroup, level-2 forward from 128 columns, level-2 inverse from 192) against the tile kernels (no_stream):
DTCWT J = 2 forward / inverse, ScatLayer forward + backward, ScatLayerj2 inference.  usage: python tools/gpu_narrow_dtcwt_fuzz.py [n]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(77)
bad, strip_runs = [], 0
for case in range(n):
    dtype = torch.float16 if rng.rand() < 0.3 else torch.float32
    H, W = 4 * int(rng.randint(8, 50)), 4 * int(rng.randint(16, 66))
    if rng.rand() < 0.3:
        H, W = 8 * (H // 8 + 1), 8 * (W // 8 + 1)
    N, C = int(rng.randint(40, 90)), int(rng.randint(1, 5))
    x = torch.randn(N, C, H, W, device=dev).to(dtype)
    xfm, ifm, sl = (m.to(dev).to(dtype) for m in (pw.DTCWTForward(J=2), pw.DTCWTInverse(), pw.ScatLayer()))
    s2 = pw.ScatLayerj2().to(dev).to(dtype)
    out, names = {}, {}
    for ns in (0, 1):
        ops.set_option('no_stream', ns)
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        y = ifm((yl, yh))
        xg = x.clone().requires_grad_(True)
        z = sl(xg)
        g, = torch.autograd.grad(z, xg, torch.ones_like(z))
        with torch.no_grad():
            z2 = s2(x)
        out[ns] = [yl, *yh, y, z.detach(), g, z2]
        names[ns] = pw.kernels_since(c0)
    ops.set_option('no_stream', 0)
    strip_runs += sum('Strip' in k for k in names[0])
    tol = 6e-3 if dtype == torch.float16 else 5e-6
    for i, (u, v) in enumerate(zip(out[0], out[1])):
        e = float((u.float() - v.float()).abs().max()) / max(float(v.float().abs().max()), 1e-30)
        if u.shape != v.shape or not e <= tol:
            bad.append((case, (N, C, H, W), str(dtype), i, e))
print(json.dumps({'cases': n, 'streaming_launches': strip_runs, 'bad': bad[:10], 'nbad': len(bad)}))
