"""Round 5: the last synthesis level of a 1024-wide pyramid - the fused kernel's one-level form (the ladder's choice) against the strip kernel."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
from pytorch_wavelets_amd.dwt import lowlevel as ll_
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=40):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
for wave in ('db4', 'db2', 'db6'):
    for shape in ((64, 3, 1024, 1024), (16, 3, 1024, 1024), (128, 3, 768, 768), (128, 3, 640, 640)):
        x = torch.randn(*shape, device=dev)
        f = pw.DWTForward(J=1, wave=wave, mode='symmetric').to(dev)
        i = pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
        with torch.no_grad():
            yl, yh = f(x)
        row = {'wave': wave, 'shape': shape}
        for fused in (True, False):
            ll_.FUSED_LEVELS = fused
            with torch.no_grad():
                i((yl, yh)); k = pw.last_kernel()
            row['fused' if fused else 'levels'] = [t(lambda: i((yl, yh))), k.split('<')[0][2:]]
        ll_.FUSED_LEVELS = True
        print(json.dumps(row), flush=True)
