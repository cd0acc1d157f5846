"""Is the module path CPU-bound?  GPU time (events) vs host launch time (no sync) per call, module vs ops level."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def both(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): f()
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    return {'gpu_ms': round(e0.elapsed_time(e1) / n, 4), 'host_launch_ms': round((t1 - t0) / n * 1e3, 4)}


res = {}
with torch.no_grad():
    x = torch.randn(128, 3, 512, 512, device=dev)
    xfm = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
    ifm = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
    yl, yh = xfm(x)
    res['fwd_module'] = both(lambda: xfm(x))
    res['inv_module'] = both(lambda: ifm((yl, yh)))
    g = (ifm.g0_col, ifm.g1_col, ifm.g0_row, ifm.g1_row)
    h = (xfm.h0_col, xfm.h1_col, xfm.h0_row, xfm.h1_row)
    res['fwd_ops'] = both(lambda: ops.afb2d_fused(x, *h, 1, 3))
    res['inv_ops'] = both(lambda: ops.sfb2d_fused(yl, yh, *g, 1))
    res['inv_ops_s1'] = both(lambda: ops.sfb2d_fused(yl, yh, *g, 1, strips=1))
    res['inv_ops_s2'] = both(lambda: ops.sfb2d_fused(yl, yh, *g, 1, strips=2))
print(json.dumps(res))
