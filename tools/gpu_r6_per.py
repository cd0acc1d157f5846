"""Round 6: periodization on the fused analysis kernel (several levels per launch; L % 4 == 0 at all) against the level-by-level
ladder of round 5 in the SAME process (ops.ROWS_PER on / off)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=50):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
CASES = [((128, 3, 512, 512), 'db4', 3, torch.float32), ((128, 3, 512, 512), 'db4', 2, torch.float32), ((128, 3, 512, 512), 'db4', 1, torch.float32),
         ((128, 3, 512, 512), 'db2', 3, torch.float32), ((128, 3, 512, 512), 'haar', 3, torch.float32), ((128, 3, 512, 512), 'db3', 3, torch.float32),
         ((128, 3, 512, 512), 'db6', 3, torch.float32), ((128, 3, 512, 512), 'db8', 3, torch.float32), ((128, 3, 512, 512), 'db8', 2, torch.float32),
         ((128, 3, 512, 512), 'db4', 3, torch.float16), ((256, 3, 256, 256), 'db4', 3, torch.float32), ((128, 3, 224, 224), 'db4', 3, torch.float32),
         ((128, 3, 640, 640), 'db4', 3, torch.float32), ((64, 3, 1024, 1024), 'db4', 3, torch.float32), ((32, 16, 2048, 2048), 'db8', 4, torch.float16)]
if len(sys.argv) > 1 and sys.argv[1] == 'deep':   # the levels below a strip-kernel level 1: where does the fused kernel stop paying?
    CASES = [((512, 1, 512, 512), 'db8', 2, torch.float16), ((512, 1, 512, 512), 'db4', 2, torch.float16), ((512, 1, 512, 512), 'db8', 2, torch.float32),
             ((512, 1, 512, 512), 'db6', 2, torch.float16), ((512, 1, 512, 512), 'db5', 2, torch.float16), ((512, 1, 512, 512), 'db7', 2, torch.float16),
             ((512, 1, 512, 512), 'db10', 2, torch.float16), ((512, 1, 256, 256), 'db8', 2, torch.float16), ((2048, 1, 256, 256), 'db8', 2, torch.float16),
             ((512, 1, 512, 512), 'db8', 3, torch.float16), ((192, 1, 512, 512), 'db8', 2, torch.float16), ((512, 1, 512, 512), 'db8', 1, torch.float16)]
for shape, wave, J, dt in CASES:
    x = torch.randn(*shape, device=dev).to(dt)
    f = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev).to(dt)
    L = f.h0_col.numel()
    b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], J, L, x.element_size(), periodization=True)
    row = {'shape': shape, 'wave': wave, 'J': J, 'dtype': str(dt)[6:]}
    for flag in (False, True):
        ops.ROWS_PER = flag
        ops._FUSED_DECLINED.clear()
        with torch.no_grad():
            f(x)
            c0 = pw.launch_count(); f(x); ks = [k for k in pw.kernels_since(c0)]
        ms = t(lambda: f(x))
        tag = 'fused' if flag else 'r5'
        row[tag + '_ms'] = ms; row[tag + '_frac'] = round(b / ms / 1e6 / 8000, 3); row[tag + '_k'] = [k.split('(')[0].strip() if k.endswith(')') else k for k in ks if not k.endswith(')')]
    print(json.dumps(row), flush=True)
