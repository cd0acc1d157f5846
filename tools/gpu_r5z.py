"""Round 5: DTCWT J = 2 / 3 on 224-wide planes - the per-level launches the policy picks (W < 256) against the fused levels-1+2 / 2+1 kernels (forced)."""
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
for shape in ((128, 3, 224, 224), (64, 3, 224, 224), (256, 3, 224, 224), (128, 3, 192, 192), (128, 3, 160, 160), (128, 3, 128, 128), (512, 3, 128, 128)):
    x = torch.randn(*shape, device=dev)
    for J in (2, 3):
        f = pw.DTCWTForward(J=J).to(dev); i = pw.DTCWTInverse().to(dev)
        row = {'shape': shape, 'J': J}
        for force in (False, True):
            ops.STREAM_FORCE = force
            ops._FUSED_DECLINED.clear()
            with torch.no_grad():
                c = f(x)
                c0 = pw.launch_count(); f(x); kf = pw.kernels_since(c0)
                c0 = pw.launch_count(); i(c); ki = pw.kernels_since(c0)
            row['forced' if force else 'policy'] = [t(lambda: f(x)), t(lambda: i(c)), [k.split('<')[0][4:] for k in kf], [k.split('<')[0][4:] for k in ki]]
        ops.STREAM_FORCE = False
        print(json.dumps(row), flush=True)
