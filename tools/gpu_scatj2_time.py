"""ScatLayerj2 inference on 64x3x256x256 (and 32x3x512x512): the in-place path (three launches writing into the 49-entry output)
against the chain + torch.cat; kernels and fraction of the HBM roofline at 16.25 B per input pixel (read x, write Z)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.scatternet import lowlevel as sl
dev = 'cuda:0'; sync = torch.cuda.synchronize
for shape in ((64, 3, 256, 256), (32, 3, 512, 512), (256, 3, 64, 64), (1024, 3, 32, 32)):
    x = torch.randn(*shape, device=dev)
    m = pw.ScatLayerj2().to(dev)
    with torch.no_grad():
        res = {}
        for fused in (True, False):
            sl.FUSED_J2 = fused
            m(x)
            c0 = pw.launch_count(); m(x); ks = pw.kernels_since(c0)
            res[fused] = (bench.time_seq_fn(lambda: m(x), 20, sync), [k.split('<')[0] + k[k.index('<'):][-12:] for k in ks])
        sl.FUSED_J2 = True
    b = 16.25 * x.numel()
    print(shape, 'in place %.4f ms (%.3f) %s | chain %.4f ms (%.3f) %s' % (res[True][0], b / res[True][0] / 8e9, res[True][1],
                                                                          res[False][0], b / res[False][0] / 8e9, res[False][1]), flush=True)
