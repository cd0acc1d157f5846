"""GPU probe: the single-axis primitives (DWT1D forward / inverse, SWT, the _rot ScatLayer that is built from them)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = {'lib': os.environ.get('WL_LIB', '')}
with torch.no_grad():
    x1 = torch.randn(64, 16, 65536, device=dev)
    d1, i1 = pw.DWT1DForward(J=3, wave='db4', mode='symmetric').to(dev), pw.DWT1DInverse(wave='db4', mode='symmetric').to(dev)
    c = d1(x1)
    out['dwt1d_fwd_ms'] = round(timeit(lambda: d1(x1)), 4)
    out['dwt1d_inv_ms'] = round(timeit(lambda: i1(c)), 4)
    out['dwt1d_rt'] = float((i1(c) - x1).abs().max())
    from pytorch_wavelets_amd.dwt.transform2d import SWTForward
    xw = torch.randn(16, 3, 512, 512, device=dev)
    sw = SWTForward(J=2, wave='db2', mode='periodic').to(dev)
    out['swt_ms'] = round(timeit(lambda: sw(xw)), 4)
    xs = torch.randn(64, 3, 256, 256, device=dev)
    sr = pw.ScatLayer(biort='near_sym_b_bp').to(dev)
    out['scat_rot_ms'] = round(timeit(lambda: sr(xs)), 4)
print(json.dumps(out))
