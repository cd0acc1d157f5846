"""Round 5: 128x3x224x224 (ImageNet crops) J = 1..3 forward / inverse - which kernels, how long; the synthesis with the lattice hint on / off."""
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
for shape in ((128, 3, 224, 224), (256, 3, 224, 224), (128, 3, 256, 256), (64, 3, 224, 224), (128, 3, 160, 160), (128, 3, 128, 128), (96, 3, 299, 299), (512, 3, 224, 224), (1024, 3, 112, 112), (512, 3, 128, 128)):
    for J in (1, 2, 3):
        x = torch.randn(*shape, device=dev)
        f = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev); i = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
        with torch.no_grad():
            c = f(x)
            c0 = pw.launch_count(); f(x); kf = pw.kernels_since(c0); g = ops._backend().wl_last_grid()
            c0 = pw.launch_count(); i(c); ki = pw.kernels_since(c0); gi = ops._backend().wl_last_grid()
        b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], J, 8, 4)
        tf, ti = t(lambda: f(x)), t(lambda: i(c))
        row = {'shape': shape, 'J': J, 'fwd_ms': tf, 'fwd_frac': round(b / tf / 1e6 / 8000, 3), 'inv_ms': ti, 'inv_frac': round(b / ti / 1e6 / 8000, 3), 'kf': kf, 'gf': g, 'ki': ki, 'gi': gi}
        prev = ops.LATTICE_MIN_ELEMS
        ops.LATTICE_MIN_ELEMS = 0
        with torch.no_grad():
            i(c)
        row['inv_ms_lattice'] = t(lambda: i(c))
        ops.LATTICE_MIN_ELEMS = prev
        print(json.dumps(row), flush=True)
