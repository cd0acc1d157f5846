"""The rotationally symmetric variants (biort 'near_sym_b_bp'): ScatLayer / ScatLayerj2 forward, inference and training step,
with level 1 in one launch (wl_dtcwt_fwd_level1_rot) against the single-axis path.  usage: python tools/gpu_rot_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dtcwt import transform_funcs as tf

dev = 'cuda:0'
a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        for _ in range(20):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return sorted(res)[1]


x = torch.randn(64, 3, 256, 256, device=dev)
for name, m in (('ScatLayer rot', pw.ScatLayer(biort='near_sym_b_bp').to(dev)),
                ('ScatLayerj2 rot', pw.ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_b_bp').to(dev)),
                ('ScatLayer plain', pw.ScatLayer().to(dev)), ('ScatLayerj2 plain', pw.ScatLayerj2().to(dev))):
    out = []
    for fused in (True, False):
        tf.FUSED_ROT = fused
        with torch.no_grad():
            c0 = pw.launch_count(); m(x); k = pw.kernels_since(c0)
            ti = timeit(lambda: m(x))

        def step():
            xg = x.detach().requires_grad_(True)
            z = m(xg)
            return torch.autograd.grad(z.sum(), xg)
        tt = timeit(step, 10)
        out.append((ti, tt, len(k), k[0] if k else ''))
    tf.FUSED_ROT = True
    by = 11 * x.numel()
    print('%-18s 64x3x256x256: inference %.4f ms (%.3f of 8 TB/s at 11 B/px) [%d launches, %s]  fwd+bwd %.4f ms   | single-axis path: %.4f ms [%d launches], fwd+bwd %.4f ms' % (
        name, out[0][0], by / out[0][0] / 1e6 / 8000, out[0][2], out[0][3], out[0][1], out[1][0], out[1][2], out[1][1]), flush=True)
