"""GPU fuzz: the DTCWT / ScatLayer streaming kernels (fused levels 1+2, lean level 1, ScatLayer, level-1 and level-2 inverse
strips) against the tile kernels (wl_set_option no_stream) on random shapes around strip / segment / halo boundaries."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.get()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
seen = set()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    W = int(rng.choice([256, 260, 264, 384, 500, 512, 516, 768, 1000, 1024, 1028, 1536, 2048]))
    H = int(rng.choice([32, 36, 64, 100, 128, 132, 256, 260, 512]))
    planes = int(max(256 // max(H // 64, 1), 8) * rng.choice([1, 1, 2]))
    C = int(rng.choice([1, 2, 3]))
    N = max(planes // C, 1)
    J = int(rng.choice([1, 2, 3]))
    dt = torch.float16 if rng.rand() < 0.25 else torch.float32
    biort = str(rng.choice(['near_sym_a', 'near_sym_a', 'legall']))
    qshift = str(rng.choice(['qshift_a', 'qshift_a', 'qshift_b', 'qshift_06']))
    x = torch.randn(N, C, H, W, device=dev).to(dt)
    xfm = pw.DTCWTForward(J=J, biort=biort, qshift=qshift).to(dev).to(dt)
    ifm = pw.DTCWTInverse(biort=biort, qshift=qshift).to(dev).to(dt)
    sl = pw.ScatLayer(biort=biort).to(dev).to(dt)
    res = {}
    kern = set()
    for ns in (0, 1):
        lib.wl_set_option(b'no_stream', ns)
        c0 = pw.launch_count()
        yl, yh = xfm(x)
        if ns == 0:
            kern.update(pw.kernels_since(c0))
        coefs = (yl + 0.1, [h * 1.1 for h in yh])
        c0 = pw.launch_count()
        rec = ifm(coefs)
        if ns == 0:
            kern.update(pw.kernels_since(c0))
        c0 = pw.launch_count()
        z = sl(x)
        if ns == 0:
            kern.update(pw.kernels_since(c0))
        res[ns] = [yl] + list(yh) + [rec, z]
    lib.wl_set_option(b'no_stream', 0)
    tol = 6e-3 if dt == torch.float16 else 1e-5
    err = max(float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)) for a, b in zip(res[0], res[1]))
    nan = any(bool(torch.isnan(a.float()).any()) for a in res[0])
    seen.update(k.split('<')[0] for k in kern)
    ok = err <= tol and not nan
    bad += not ok
    print(json.dumps({'ok': ok, 'shape': [N, C, H, W], 'J': J, 'dtype': str(dt).split('.')[-1], 'biort': biort, 'qshift': qshift,
                      'err': err, 'kernels': sorted(kern)}), flush=True)
print('kernels seen:', sorted(seen))
print('FAILURES: %d' % bad)
