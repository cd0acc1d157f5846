"""Round 6: periodization on the fused SYNTHESIS kernel against the level-by-level ladder in the SAME process (ops.IROWS_PER on / off)."""
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
         ((128, 3, 512, 512), 'db5', 3, torch.float32), ((128, 3, 512, 512), 'db6', 3, torch.float32), ((128, 3, 512, 512), 'db8', 3, torch.float32),
         ((128, 3, 512, 512), 'db9', 3, torch.float32), ((128, 3, 512, 512), 'db10', 3, torch.float32), ((128, 3, 512, 512), 'bior4.4', 3, torch.float32),
         ((128, 3, 512, 512), 'db4', 3, torch.float16), ((128, 3, 512, 512), 'haar', 3, torch.float16), ((128, 3, 512, 512), 'db2', 3, torch.float16),
         ((256, 3, 256, 256), 'db4', 3, torch.float32), ((128, 3, 224, 224), 'db4', 3, torch.float32), ((512, 3, 128, 128), 'db4', 2, torch.float32),
         ((128, 3, 640, 640), 'db4', 3, torch.float32), ((64, 3, 1024, 1024), 'db4', 3, torch.float32), ((32, 16, 2048, 2048), 'db8', 4, torch.float16)]
if len(sys.argv) > 1 and sys.argv[1] == 'narrow':
    ops_policy_off = True
    CASES = [((n, 3, w, w), wv, J, torch.float32) for (n, w) in ((128, 224), (256, 224), (128, 256), (256, 256), (512, 128), (1024, 112), (128, 320), (128, 288), (512, 64)) for wv, J in (('db4', 3), ('db2', 2), ('haar', 1), ('db4', 1))]
    CASES += [((48, 3, 512, 512), 'db4', 3, torch.float32), ((64, 3, 512, 512), 'db4', 3, torch.float32), ((96, 3, 384, 384), 'db4', 3, torch.float32)]
if len(sys.argv) > 1 and sys.argv[1] == 'widths':
    CASES = [((128, 3, w, w), wv, J, dt) for w in (256, 320, 384, 448, 576, 768) for wv, J, dt in (('db4', 3, torch.float32), ('db2', 2, torch.float32), ('haar', 3, torch.float16), ('db2', 3, torch.float16))]
    CASES += [((128, 3, 512, 512), 'db7', 3, torch.float32), ((128, 3, 512, 512), 'db3', 3, torch.float16), ((48, 3, 512, 512), 'db4', 3, torch.float32), ((512, 3, 512, 512), 'db4', 3, torch.float32)]
for shape, wave, J, dt in CASES:
    x = torch.randn(*shape, device=dev).to(dt)
    f = pw.DWTForward(J=J, wave=wave, mode='periodization').to(dev).to(dt)
    i = pw.DWTInverse(wave=wave, mode='periodization').to(dev).to(dt)
    L = f.h0_col.numel()
    b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], J, L, x.element_size(), periodization=True)
    row = {'shape': shape, 'wave': wave, 'J': J, 'dtype': str(dt)[6:]}
    with torch.no_grad():
        c = f(x)
    for flag in (False, True):
        ops.IROWS_PER = flag
        ops.IROWS_PER_POLICY = not (len(sys.argv) > 1 and sys.argv[1] == 'narrow')
        ops._FUSED_DECLINED.clear()
        with torch.no_grad():
            i(c)
            c0 = pw.launch_count(); r = i(c); ks = pw.kernels_since(c0)
        ms = t(lambda: i(c))
        tag = 'fused' if flag else 'r5'
        row[tag + '_ms'] = ms; row[tag + '_frac'] = round(b / ms / 1e6 / 8000, 3); row[tag + '_k'] = [k for k in ks if not k.endswith(')')]
        row[tag + '_rt'] = float((r - x).abs().max() / x.abs().max())
    print(json.dumps(row), flush=True)
