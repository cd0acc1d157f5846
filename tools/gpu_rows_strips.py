"""Streaming analysis kernel: whole planes vs automatic cutting vs all planes cut (HIP-event timed)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0, h1 = filters.dwt_analysis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
res = {}
for planes in [int(v) for v in os.environ.get('SWEEP_PLANES', '384').split(',')]:
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for J in (1, 3):
        for strips in (1, 0, 2):
            f = lambda: ops.afb2d_fused(x, *th, 1, J, strips=strips)
            for _ in range(3): f()
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res['p%d_J%d_s%d' % (planes, J, strips)] = round(e0.elapsed_time(e1) / 20, 4)
print(json.dumps(res))
