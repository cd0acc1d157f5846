"""GPU probe: a large batch (64-bit plane offsets: > 2^31 elements per tensor) through the DTCWT streaming kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.get()
x = torch.randn(300, 3, 1024, 1024, device=dev)
xfm, ifm, sl = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev), pw.ScatLayer().to(dev)
with torch.no_grad():
    c0 = pw.launch_count()
    yl, yh = xfm(x)
    k = pw.kernels_since(c0)
    rec = ifm((yl, yh))
    z = sl(x)
    print(json.dumps({'numel_highs1': yh[0].numel(), 'kernels': k, 'roundtrip': float((rec - x).abs().max() / x.abs().max())}))
    # the last planes against the tile kernels on a small slice (same values whatever the batch)
    xs = x[-2:].clone()
    lib.wl_set_option(b'no_stream', 1)
    yl2, yh2 = xfm(xs)
    z2 = sl(xs)
    lib.wl_set_option(b'no_stream', 0)
    errs = [float((yl[-2:] - yl2).abs().max() / yl2.abs().max())] + [float((a[-2:] - b).abs().max() / b.abs().max()) for a, b in zip(yh, yh2)] + [float((z[-2:] - z2).abs().max() / z2.abs().max())]
    print(json.dumps({'last_planes_vs_tile': max(errs)}))
