"""Row segments of the lean level-1 / level-2 kernels (A/B build with -DWL_SEG_PROBE; WL_SEG_N forces the number of segments):
the three launches of ScatLayerj2 on 64x3x256x256 and the ScatLayer of config 4."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
m = pw.ScatLayerj2().to(dev)
out = {'seg_n': os.environ.get('WL_SEG_N')}
with torch.no_grad():
    for name, shape in (('first 192x256x256', (64, 3, 256, 256)), ('second-order 1152x128x128', (64, 18, 128, 128)), ('cfg4 768x256x256', (256, 3, 256, 256))):
        x = torch.randn(*shape, device=dev)
        n, c, H, W = shape
        z = torch.empty((n, 7, c, H // 2, W // 2), device=dev)
        q = (H // 2) * (W // 2)
        f = lambda: ops.scat_fwd1_into(x, z, 7 * c * q, 0, c * q, m.h0o, m.h1o, m.mode, 0.01)
        f(); k = pw.last_kernel()
        out[name] = (round(bench.time_seq_fn(f, 30, sync), 4), k[-12:])
    x = torch.randn(64, 3, 256, 256, device=dev)
    z = torch.empty((64, 49, 3, 64, 64), device=dev)
    f = lambda: ops.scat_fwd2_into(x, z, 49 * 3 * 4096, 0, 7 * 3 * 4096, m.h0a, m.h0b, m.h1a, m.h1b, 0.01)
    f(); k = pw.last_kernel()
    out['level 2 192x256x256'] = (round(bench.time_seq_fn(f, 30, sync), 4), k[-12:])
print(json.dumps(out))
