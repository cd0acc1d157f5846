"""In-kernel cycle breakdown of the strip kernel (timing builds: WL_STRIP_ABLATE & 8), config-5 level 1."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
h0, h1 = filters.dwt_analysis_taps('db8')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
x = torch.randn(32, 16, 2048, 2048, device=dev, dtype=torch.float16)
for _ in range(3):
    ll, hh = ops.afb2d_stream(x, *th, 2, force=True)
torch.cuda.synchronize()
v = ll.reshape(-1, 1024 * 1024)[:, :12].double().mean(0) * 1024
print(json.dumps({'lib': os.environ.get('WL_LIB'), 'compute_cycles': {'barrier': int(v[0]), 'work': int(v[1])},
                  'stager_cycles': {'vmwait': int(v[8]), 'stage': int(v[9]), 'barrier': int(v[10]), 'issue': int(v[11])}}))
