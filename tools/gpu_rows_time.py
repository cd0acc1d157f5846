"""In-kernel cycle breakdown of the streaming kernel (timing builds: WL_ROWS_ABLATE & 8)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
h0, h1 = filters.dwt_analysis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
for planes in (128, 384):
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for J in (1, 3):
        for _ in range(3):
            yl, yh = ops.afb2d_fused(x, *th, 1, J, strips=1)
        torch.cuda.synchronize()
        v = yl[:, 0, 0, :11].double().mean(0) * 64
        print(json.dumps({'lib': os.environ.get('WL_LIB'), 'planes': planes, 'J': J,
                          'L1wave_cycles': {'barrier': int(v[0]), 'feeds': int(v[1]), 'sched': int(v[2])},
                          'loader_cycles': {'vmwait': int(v[8]), 'barrier': int(v[9]), 'issue': int(v[10])}}))
