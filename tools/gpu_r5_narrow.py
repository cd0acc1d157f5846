"""Round 5: the strip kernels on levels NARROWER than the engine's float32 policy (rows of 2 KiB analysis / 1 KiB synthesis) - with
the lattice column pass the long-filter strip kernel may beat the tile kernels there.  Forced strips (ops.STREAM_FORCE) against
the default dispatch, same box, per wavelet / shape.  usage: python tools/gpu_r5_narrow.py"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=30):
    with torch.no_grad():
        fn(); fn(); c0 = pw.launch_count(); fn(); ks = [k.replace('float', 'f') for k in pw.kernels_since(c0) if not k.endswith(')')]
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(3)), 4), ks
for wave, shape, J, mode in (('db8', (128, 3, 512, 512), 3, 'symmetric'), ('sym8', (128, 3, 512, 512), 3, 'periodization'), ('db6', (64, 3, 1024, 1024), 3, 'symmetric'),
                             ('db4', (64, 3, 1024, 1024), 3, 'symmetric'), ('db4', (16, 3, 1024, 1024), 3, 'symmetric'), ('db8', (32, 3, 1024, 1024), 4, 'symmetric'),
                             ('db7', (128, 3, 512, 512), 3, 'symmetric'), ('db10', (128, 3, 512, 512), 2, 'symmetric')):
    x = torch.randn(*shape, device=dev)
    m = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev); im = pw.DWTInverse(wave=wave, mode=mode).to(dev)
    yl, yh = m(x)
    out = {'case': '%s %s J%d %s' % (wave, 'x'.join(map(str, shape)), J, mode)}
    for force in (False, True):
        ops.STREAM_FORCE = force
        f, kf = t(lambda: m(x)); i, ki = t(lambda: im((yl, yh)))
        out['fwd_force' if force else 'fwd'] = f; out['inv_force' if force else 'inv'] = i
        out['k_force' if force else 'k'] = [kf, ki]
    ops.STREAM_FORCE = False
    print(json.dumps(out), flush=True)
    del x, yl, yh
