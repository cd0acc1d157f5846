"""In-kernel cycle breakdown of the streaming synthesis kernel (timing builds: WL_IROWS_ABLATE & 8): per wave, the
cycles spent waiting at the barrier and working (compute waves) / waiting for DMA, at the barrier, issuing (loaders)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
h0, h1 = filters.dwt_analysis_taps('db4')
g0, g1 = filters.dwt_synthesis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
for planes in (512,):
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for J in (1, 3):
        yl, yh = x, []
        for _ in range(J):
            yl, hi = ops.afb2d(yl, *th, 1)
            yh.append(hi)
        for _ in range(3):
            y = ops.sfb2d_fused(yl, yh, *tg, 1, strips=1)
        torch.cuda.synchronize()
        v = (y[:, 0, 0, :56].double().mean(0) * 64).reshape(14, 4)[:, :3].round().long().tolist()
        print(json.dumps({'lib': os.environ.get('WL_LIB'), 'planes': planes, 'J': J, 'waves': v}))
