"""Round 5: few planes on narrow levels - strip kernel (forced; planes packed as far as the chip stays full) against the policy's
choice, J = 1 periodization, 384 planes (the metric's batch) and 96."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=60):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
be = ops._backend()
for dt in (torch.float32, torch.float16):
    for wave in ('db4', 'db8'):
        for planes in (384, 96, 1536):
            for W in (128, 256, 512):
                x = torch.randn(planes // 3, 3, W, W, device=dev).to(dt)
                f = pw.DWTForward(J=1, wave=wave, mode='periodization').to(dev).to(dt)
                i = pw.DWTInverse(wave=wave, mode='periodization').to(dev).to(dt)
                row = {'dtype': str(dt)[6:], 'wave': wave, 'W': W, 'planes': planes}
                for force in (False, True):
                    ops.STREAM_FORCE = force
                    with torch.no_grad():
                        yl, yh = f(x); kf = pw.last_kernel(); gf = be.wl_last_grid(); i((yl, yh)); ki = pw.last_kernel(); gi = be.wl_last_grid()
                    tag = 'strip' if force else 'default'
                    row[tag + '_fwd'] = t(lambda: f(x)); row[tag + '_inv'] = t(lambda: i((yl, yh)))
                    row[tag + '_k'] = [kf.split('<')[0][2:], gf, ki.split('<')[0][2:], gi]
                ops.STREAM_FORCE = False
                print(json.dumps(row), flush=True)
