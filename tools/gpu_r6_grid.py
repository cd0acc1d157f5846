"""Round 6: wavelet x mode x dtype grid at 128x3x512^2 J = 3 (and one wider / one narrower shape) - forward / inverse time, fraction of
the HBM roofline at the algorithmic bytes, the kernels that did the work.  Looks for policy traps: a cell far below its neighbours."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'; sync = torch.cuda.synchronize
short = lambda ks: ','.join(k.split('(')[0].strip() for k in ks if not k.endswith(')'))
WAVES = ['haar', 'db2', 'db3', 'db4', 'db5', 'db6', 'db7', 'db8', 'db9', 'db10', 'sym4', 'sym8', 'coif1', 'coif2', 'coif3',
         'bior1.3', 'bior2.2', 'bior3.3', 'bior4.4', 'bior6.8']
shapes = [(128, 3, 512, 512)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1].split('x'))]
for shape in shapes:
    for dt in (torch.float32, torch.float16):
        x = torch.randn(*shape, device=dev).to(dt)
        for wave in WAVES:
            for mode in ('symmetric', 'zero', 'reflect', 'periodization', 'periodic'):
                fx = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev).to(dt); fi = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
                with torch.no_grad():
                    c = fx(x)
                    c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
                    c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
                    tf = min(bench.time_seq_fn(lambda: fx(x), 20, sync) for _ in range(2)); ti = min(bench.time_seq_fn(lambda: fi(c), 20, sync) for _ in range(2))
                yl, yh = c
                b = x.element_size() * (x.numel() + yl.numel() + sum(h.numel() for h in yh))
                print(json.dumps({'shape': shape, 'dtype': str(dt)[6:], 'wave': wave, 'L': fx.h0_col.numel(), 'mode': mode, 'fwd_ms': round(tf, 4), 'fwd_frac': round(b / tf / 8e9, 3),
                                  'inv_ms': round(ti, 4), 'inv_frac': round(b / ti / 8e9, 3), 'kf': short(kf), 'ki': short(ki)}), flush=True)
