"""Round 5: where do the one-level strip kernels (now several planes per workgroup on narrow levels) beat what the engine's policy
picks today?  J = 1 forward / inverse, periodization (the fused multi-level kernels do not take it), default policy vs forced strip."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=40):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
for dt in (torch.float16, torch.float32):
    for wave in ('db2', 'db4', 'db8'):
        for W in (64, 128, 256, 512):
            planes = max(64, 2 ** 27 // (W * W) // 16 * 16)       # 128 M elements... bounded
            planes = min(planes, 8192)
            x = torch.randn(planes // 16, 16, W, W, device=dev).to(dt)
            f = pw.DWTForward(J=1, wave=wave, mode='periodization').to(dev).to(dt)
            i = pw.DWTInverse(wave=wave, mode='periodization').to(dev).to(dt)
            row = {'dtype': str(dt)[6:], 'wave': wave, 'W': W, 'planes': planes}
            for force in (False, True):
                ops.STREAM_FORCE = force
                with torch.no_grad():
                    yl, yh = f(x); kf = pw.last_kernel(); i((yl, yh)); ki = pw.last_kernel()
                tag = 'strip' if force else 'default'
                row[tag + '_fwd'] = t(lambda: f(x)); row[tag + '_inv'] = t(lambda: i((yl, yh)))
                row[tag + '_k'] = [kf.split('<')[0], ki.split('<')[0]]
            ops.STREAM_FORCE = False
            print(json.dumps(row), flush=True)
            del x, yl, yh
