"""Round 5: ImageNet-sized batches through ScatLayer / DTCWT - time, fraction of the roofline (11 / 20 B per pixel), kernels, grids."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=100):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
be = ops._backend()
for shape in ((128, 3, 224, 224), (64, 3, 224, 224), (256, 3, 224, 224), (128, 3, 256, 256), (32, 3, 224, 224), (128, 3, 112, 112), (128, 64, 56, 56)):
    x = torch.randn(*shape, device=dev)
    px = shape[0] * shape[1] * shape[2] * shape[3]
    s = pw.ScatLayer().to(dev)
    with torch.no_grad():
        c0 = pw.launch_count(); s(x); ks = pw.kernels_since(c0); g = be.wl_last_grid()
    ts = t(lambda: s(x))
    row = {'shape': shape, 'scat_ms': ts, 'scat_frac': round(px * 11 / ts / 1e6 / 8000, 3), 'scat_k': ks, 'grid': g}
    for J in (1, 2, 3):
        f = pw.DTCWTForward(J=J).to(dev); i = pw.DTCWTInverse().to(dev)
        with torch.no_grad():
            c = f(x)
            c0 = pw.launch_count(); f(x); kf = pw.kernels_since(c0)
            c0 = pw.launch_count(); i(c); ki = pw.kernels_since(c0)
        tf, ti = t(lambda: f(x)), t(lambda: i(c))
        row['dt%d' % J] = [tf, round(px * 20 / tf / 1e6 / 8000, 3), ti, round(px * 20 / ti / 1e6 / 8000, 3), [k.split('<')[0] for k in kf], [k.split('<')[0] for k in ki]]
    print(json.dumps(row), flush=True)
