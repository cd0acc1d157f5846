"""A few launches of the streaming kernel only (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
h0, h1 = filters.dwt_analysis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
planes = int(os.environ.get('PLANES', '384'))
J = int(os.environ.get('J', '3'))
x = torch.randn(planes, 1, 512, 512, device=dev)
for _ in range(5):
    ops.afb2d_fused(x, *th, 1, J, strips=1)
torch.cuda.synchronize()
