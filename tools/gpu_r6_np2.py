"""Round 6: three levels of a long orthogonal filter in symmetric / reflect mode - the fused analysis with exactly sized LL rings (NP2) against
the two-launch path of round 5 (build with -DWL_NO_NP2 is not needed: ops.ROWS_NP2 = False makes the Python ladder ask for two levels + one)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
short = lambda ks: ','.join(k.split('(')[0].strip() for k in ks if not k.endswith(')'))
for shape in ((128, 3, 512, 512), (128, 3, 448, 448), (64, 3, 512, 512)):
    x = torch.randn(*shape, device=dev)
    for wave in ('db6', 'db7', 'db8', 'sym8', 'coif2', 'db9', 'db10', 'db5', 'db4'):
        for mode in ('symmetric', 'reflect', 'zero'):
            fx = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev)
            with torch.no_grad():
                c = fx(x)
                c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
                tf = min(bench.time_seq_fn(lambda: fx(x), 30, sync) for _ in range(3))
            yl, yh = c
            b = 4 * (x.numel() + yl.numel() + sum(h.numel() for h in yh))
            print(json.dumps({'shape': shape, 'wave': wave, 'mode': mode, 'fwd_ms': round(tf, 4), 'fwd_frac': round(b / tf / 8e9, 3), 'k': short(kf)}), flush=True)
