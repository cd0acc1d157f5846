"""Round 5: the fused analysis kernel on rows of 2-3 KiB (three pieces per row) and on the padded odd-width ll of a strip level -
which pyramids gain?  Same process, ops.ROWS_3KIB / ops.PAD_ODD_LL on and off."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=30):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
for shape, J in (((64, 3, 1024, 1024), 3), ((64, 3, 1024, 1024), 2), ((32, 3, 2048, 2048), 3), ((16, 3, 1024, 1024), 3), ((128, 3, 640, 640), 3), ((128, 3, 640, 640), 1),
                 ((128, 3, 640, 640), 2), ((128, 3, 384, 640), 3), ((128, 3, 768, 768), 3), ((128, 3, 768, 768), 1), ((256, 3, 600, 600), 2)):
    x = torch.randn(*shape, device=dev)
    f = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev)
    row = {'shape': shape, 'J': J}
    for tag, r3, pad in (("new", True, True), ("old", False, False)):
        ops.ROWS_3KIB, ops.PAD_ODD_LL = r3, pad
        ops._FUSED_DECLINED.clear()
        with torch.no_grad():
            c0 = pw.launch_count(); f(x); ks = [k.split('<')[0][2:] + k[k.index('<'):][:14] for k in pw.kernels_since(c0) if 'armed' not in k and 'aux' not in k]
        row[tag] = [t(lambda: f(x)), ks]
    print(json.dumps(row), flush=True)
    del x
