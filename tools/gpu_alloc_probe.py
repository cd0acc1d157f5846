import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def gpu(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)
res = {}
with torch.no_grad():
    x = torch.randn(128, 3, 512, 512, device=dev)
    xfm = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
    h = (xfm.h0_col, xfm.h1_col, xfm.h0_row, xfm.h1_row)
    ptrs = []
    def f_ops():
        r = ops.afb2d_fused(x, *h, 1, 3); ptrs.append(r[1][0].data_ptr()); return r
    def f_fn():
        r = lowlevel.AFB2DMulti.apply(x, *h, 1, 3); ptrs.append(r[1].data_ptr()); return r
    def f_mod():
        r = xfm(x); ptrs.append(r[1][0].data_ptr()); return r
    for name, f in (('ops', f_ops), ('fn', f_fn), ('mod', f_mod), ('ops2', f_ops)):
        ptrs.clear()
        res[name] = gpu(f)
        res[name + '_distinct_ptrs'] = len(set(ptrs))
print(json.dumps(res))
