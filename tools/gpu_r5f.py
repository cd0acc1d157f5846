"""Round 5: (i) config 5's inverse with every level on the strip kernel (the coarsest level - 256 output columns - runs on the tile
kernel by the engine's policy), (ii) 14-tap wavelets on the default dispatch after the tile instantiations."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=20):
    with torch.no_grad():
        fn(); fn(); c0 = pw.launch_count(); fn(); ks = [k.replace('float', 'f') for k in pw.kernels_since(c0) if not k.endswith(')')]
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(3)), 4), ks
xh = torch.randn(32, 16, 2048, 2048, device=dev, dtype=torch.float16)
m5 = pw.DWTForward(J=4, wave='db8', mode='periodization').to(dev).half()
yl, yh = m5(xh); del xh
i5 = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half()
for force in (False, True, False, True):
    ops.STREAM_FORCE = force
    print(json.dumps({'cfg5_inv_force' if force else 'cfg5_inv': t(lambda: i5((yl, yh)), 5)}), flush=True)
ops.STREAM_FORCE = False
del yl, yh
x = torch.randn(128, 3, 512, 512, device=dev)
for wave in ('db7', 'sym7', 'db9', 'db8', 'db4'):
    m = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev); im = pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
    y = m(x)
    print(json.dumps({wave: [t(lambda: m(x)), t(lambda: im(y))]}), flush=True)
