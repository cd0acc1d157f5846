"""DWTInverse J=3 symmetric 128x3x512x512 float32 by tap count: the fused streaming synthesis kernel against one launch per level.
usage: python tools/gpu_inv_taps_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel as _ll

dev = 'cuda:0'
x = torch.randn(128, 3, 512, 512, device=dev)
for wave, L in (('db2', 4), ('db3', 6), ('db4', 8), ('db5', 10), ('db6', 12), ('db7', 14)):
    fx, fi = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev), pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
    with torch.no_grad():
        c = fx(x)
        out = []
        for fused in (True, False):
            _ll.FUSED_LEVELS = fused
            c0 = pw.launch_count(); fi(c); k = pw.kernels_since(c0)
            out.append((bench.time_seq_fn(lambda: fi(c), 20, torch.cuda.synchronize), k))
            c0 = pw.launch_count(); fx(x); k = pw.kernels_since(c0)
            out.append((bench.time_seq_fn(lambda: fx(x), 20, torch.cuda.synchronize), k))
        _ll.FUSED_LEVELS = True
    b = bench.algorithmic_bytes_fwd(128, 3, 512, 512, 3, L, 4)
    print('%s L=%d: inv fused %.4f ms (%.3f) %s | per level %.4f ms (%.3f) %s || fwd fused %.4f ms (%.3f) %s | per level %.4f (%.3f) %s' % (
        wave, L, out[0][0], b / out[0][0] / 8e9, out[0][1], out[2][0], b / out[2][0] / 8e9, out[2][1],
        out[1][0], b / out[1][0] / 8e9, out[1][1], out[3][0], b / out[3][0] / 8e9, out[3][1]), flush=True)
