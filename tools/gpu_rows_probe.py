"""A/B on one box: streaming multi-level analysis kernel vs one tile launch per level (HIP-event timed)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops, filters, _lib
from pytorch_wavelets_amd.dwt import lowlevel

dev = torch.device('cuda:0')
N = int(os.environ.get('PROBE_N', '128'))
x = torch.randn(N, 3, 512, 512, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


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


def alg_bytes(J, L=8):
    n, h, w = 512 * 512, 512, 512
    for _ in range(J):
        h, w = (h + L - 1) // 2, (w + L - 1) // 2
        n += 3 * h * w
    return (n + h * w) * 4 * N * 3


out = {}
with torch.no_grad():
    for J in (1, 2, 3):
        xfm = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev)
        for fused in (True, False):
            lowlevel.FUSED_LEVELS = fused
            ms = timed(lambda: xfm(x))
            name = _lib.get().wl_last_kernel().decode().split('K = ')[-1].rstrip(']')
            out['J%d_%s' % (J, 'rows' if fused else 'tile')] = {'ms': round(ms, 4), 'TBs': round(alg_bytes(J) / ms / 1e9, 3),
                                                              'frac': round(alg_bytes(J) / ms / 1e9 / 8.0, 4), 'last_kernel': name}
    lowlevel.FUSED_LEVELS = True
    y = torch.empty_like(x)
    ms = timed(lambda: y.copy_(x))
    out['copy'] = {'ms': round(ms, 4), 'TBs': round(2 * x.numel() * 4 / ms / 1e9, 3)}
print(json.dumps(out, indent=1))
