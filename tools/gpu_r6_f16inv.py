"""Round 6: float16 inverse, fused synthesis against the per-level ladder (ops.IROWS_F16_MAXL = 99 / 8) in the same process."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
short = lambda ks: ','.join(k.split('(')[0].strip() for k in ks if not k.endswith(')'))
for shape in ((128, 3, 512, 512), (128, 3, 224, 224), (64, 3, 1024, 1024)):
    x = torch.randn(*shape, device=dev).half()
    for wave in ('db4', 'db5', 'db6', 'db7', 'db8', 'db9', 'bior4.4'):
        for mode in ('symmetric', 'zero', 'periodic'):
            for J in (1, 3):
                fx = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev).half(); fi = pw.DWTInverse(wave=wave, mode=mode).to(dev).half()
                row = {'shape': shape, 'wave': wave, 'mode': mode, 'J': J}
                with torch.no_grad():
                    c = fx(x)
                    for tag, v in (('fused', 99), ('ladder', 8)):
                        ops.IROWS_F16_MAXL = v; ops._FUSED_DECLINED.clear()
                        fi(c); c0 = pw.launch_count(); fi(c); row[tag + '_k'] = short(pw.kernels_since(c0))
                        row[tag + '_ms'] = round(min(bench.time_seq_fn(lambda: fi(c), 20, sync) for _ in range(3)), 4)
                print(json.dumps(row), flush=True)
