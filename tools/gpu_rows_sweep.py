"""Streaming analysis kernel: time vs number of planes and vs nlev (one library build per process: WL_LIB selects it)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_wavelets_amd import ops, filters

dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0, h1 = filters.dwt_analysis_taps('db4')
th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {'lib': os.environ.get('WL_LIB', 'product')}
planes_list = [int(v) for v in os.environ.get('SWEEP_PLANES', '384').split(',')]
for planes in planes_list:
    x = torch.randn(planes, 1, 512, 512, device=dev)
    for J in (1, 3):
        ms = timed(lambda: ops.afb2d_fused(x, *th, 1, J, strips=1))
        res['p%d_J%d' % (planes, J)] = round(ms, 4)
print(json.dumps(res))
