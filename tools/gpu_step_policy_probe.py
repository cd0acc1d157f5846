"""The step loop (forward then inverse, free running) under different plane-cutting policies of the two streaming kernels."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
h0, h1 = filters.dwt_analysis_taps('db4'); g0, g1 = filters.dwt_synthesis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
res = {'lib': os.environ.get('WL_LIB')}
for planes in [int(v) for v in os.environ.get('SWEEP_PLANES', '384').split(',')]:
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for fs in (0, 1, 2):
        for iv in (0, 1, 2):
            def step():
                yl, yh = ops.afb2d_fused(x, *th, 1, 3, strips=fs)
                return ops.sfb2d_fused(yl, yh, *tg, 1, strips=iv)
            for _ in range(150): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): step()
            torch.cuda.synchronize()
            res['p%d_f%d_i%d' % (planes, fs, iv)] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
print(json.dumps(res))
