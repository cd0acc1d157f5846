"""Round 5: three synthesis levels of few narrow planes - one fused launch (one plane per workgroup: no waves for a second) against
the two coarse levels in one launch + the finest on its own (both packed and cut)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=100):
    with torch.no_grad():
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
for shape in ((128, 3, 224, 224), (128, 3, 256, 256), (64, 3, 224, 224), (128, 3, 160, 160), (96, 3, 299, 299), (256, 3, 224, 224), (128, 3, 512, 512)):
    for J in (3, 2):
        x = torch.randn(*shape, device=dev)
        f = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev); i = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
        with torch.no_grad():
            yl, yh = f(x)
            def split():
                ll = i((yl, yh[1:]))
                h = yh[0]
                return i((ll[..., :h.shape[-2], :h.shape[-1]], [h]))
            assert float((split() - i((yl, yh))).abs().max()) < 1e-4
        print(json.dumps({'shape': shape, 'J': J, 'one_launch': t(lambda: i((yl, yh))), 'coarse_then_fine': t(split)}), flush=True)
