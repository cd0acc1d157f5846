"""GPU fuzz, round 6 (late): the 13 / 19-tap level-1 pair on the streaming kernels - DTCWT forward / inverse / gradient with near_sym_b and
qshift_b / qshift_d / qshift_a, ScatLayer(near_sym_b) inference + training step, ScatLayer(near_sym_b_bp) inference (lean MODE 6) -
against the same transforms on the tile kernels (wl_set_option no_stream) on random shapes around strip / segment / pair-of-planes
boundaries.  usage: gpu_round6c_fuzz.py [seed] [cases]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = torch.device('cuda:0')
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
seen = set()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    W = int(rng.choice([128, 132, 200, 224, 256, 260, 384, 500, 512, 516, 768, 1000, 1024, 1028, 1536]))
    H = int(rng.choice([40, 44, 64, 100, 128, 132, 224, 256, 260, 512]))
    planes = int(max(256 // max(H // 64, 1), 8) * rng.choice([1, 1, 2]))
    C = int(rng.choice([1, 2, 3]))
    N = max(planes // C, 1)
    J = int(rng.choice([1, 2, 3]))
    dt = torch.float16 if rng.rand() < 0.25 else torch.float32
    qshift = str(rng.choice(['qshift_b', 'qshift_b', 'qshift_d', 'qshift_a']))
    x = torch.randn(N, C, H, W, device=dev).to(dt)
    xfm = pw.DTCWTForward(J=J, biort='near_sym_b', qshift=qshift).to(dev).to(dt)
    ifm = pw.DTCWTInverse(biort='near_sym_b', qshift=qshift).to(dev).to(dt)
    sl = pw.ScatLayer(biort='near_sym_b').to(dev).to(dt)
    sr = pw.ScatLayer(biort='near_sym_b_bp').to(dev).to(dt)
    res, kern = {}, set()
    for ns in (0, 1):
        ops.set_option('no_stream', ns)
        try:
            c0 = pw.launch_count()
            xg = x.clone().requires_grad_(True)
            yl, yh = xfm(xg)
            g, = torch.autograd.grad([yl] + list(yh), xg, [yl.detach() * 0.5 + 0.1] + [h.detach() * 1.1 for h in yh])
            with torch.no_grad():
                rec = ifm((yl.detach() + 0.1, [h.detach() * 1.1 for h in yh]))
                zr = sr(x)
            xs = x.clone().requires_grad_(True)
            z = sl(xs)
            gz, = torch.autograd.grad(z, xs, z.detach())
            if ns == 0:
                kern.update(pw.kernels_since(c0))
        finally:
            ops.set_option('no_stream', 0)
        res[ns] = [yl.detach()] + [h.detach() for h in yh] + [g, rec, z.detach(), gz, zr]
    tol = 8e-3 if dt == torch.float16 else 1e-5
    err = max(float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)) for a, b in zip(res[0], res[1]))
    nan = any(bool(torch.isnan(a.float()).any()) for a in res[0])
    seen.update(k[:k.index('>') + 1] if '>' in k else k for k in kern if 'Strip' in k)
    ok = err <= tol and not nan
    bad += not ok
    print(json.dumps({'ok': ok, 'shape': [N, C, H, W], 'J': J, 'dtype': str(dt).split('.')[-1], 'qshift': qshift, 'err': err,
                      'kernels': sorted(k.replace('float', 'f') for k in kern if 'Strip' in k)}), flush=True)
print('streaming kernels seen:', sorted(seen))
print('FAILURES: %d' % bad)
