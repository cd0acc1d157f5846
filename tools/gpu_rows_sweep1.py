"""Streaming analysis kernel, J=1 only: time vs number of planes (WL_LIB selects the build)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0, h1 = filters.dwt_analysis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
res = {'lib': os.environ.get('WL_LIB', 'product')}
for planes in [int(v) for v in os.environ.get('SWEEP_PLANES', '384').split(',')]:
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for _ in range(3):
        ops.afb2d_fused(x, *th, 1, 1, strips=1)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.afb2d_fused(x, *th, 1, 1, strips=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res['p%d' % planes] = [round(ms, 4), round(ms / planes * 384, 4)]
    del x
print(json.dumps(res))
