"""Round 6: DTCWT / ScatLayer with the 13 / 19-tap level-1 pair (near_sym_b) - the streaming level-1 kernels (lean forward
WlDtFwd12Strip<T, 13, 19, ..>, streaming inverse WlDtInv1Strip<T, 19, 13>) against the tile kernels they replace
(wl_set_option no_stream: the whole transform on tile kernels; 'l1tile': only level 1 - by WL_NSB_OFF in the environment of a second
process), same process, same tensors.  One JSON line per case."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
short = lambda ks: ','.join(k.split('(')[0].strip().replace('float', 'f') for k in ks if not k.endswith(')'))


def dt(shape, J, biort, qshift, dtype=torch.float32):
    x = torch.randn(*shape, device=dev).to(dtype)
    fx = pw.DTCWTForward(J=J, biort=biort, qshift=qshift).to(dev).to(dtype)
    fi = pw.DTCWTInverse(biort=biort, qshift=qshift).to(dev).to(dtype)
    rec = {'case': 'dtcwt', 'shape': shape, 'J': J, 'biort': biort, 'qshift': qshift, 'dtype': str(dtype).split('.')[-1]}
    for ns in (0, 1):
        ops.set_option('no_stream', ns)
        try:
            with torch.no_grad():
                c = fx(x)
                c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
                c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
                tf = min(bench.time_seq_fn(lambda: fx(x), 30, sync) for _ in range(3))
                ti = min(bench.time_seq_fn(lambda: fi(c), 30, sync) for _ in range(3))
            xg = x.clone().requires_grad_(True)
            def step():
                yl, yh = fx(xg)
                torch.autograd.grad([yl] + list(yh), xg, [yl] + list(yh))
            tb = min(bench.time_seq_fn(step, 20, sync) for _ in range(2)) if dtype == torch.float32 else 0.0
        finally:
            ops.set_option('no_stream', 0)
        yl, yh = c
        b = x.element_size() * (x.numel() + yl.numel() + sum(h.numel() for h in yh))
        tag = 'tile' if ns else 'stream'
        rec.update({tag + '_fwd_ms': round(tf, 4), tag + '_inv_ms': round(ti, 4), tag + '_fwdbwd_ms': round(tb, 4),
                    tag + '_fwd_frac': round(b / tf / 8e9, 3), tag + '_inv_frac': round(b / ti / 8e9, 3), tag + '_k': short(kf) + ' | ' + short(ki)})
    print(json.dumps(rec), flush=True)


def scat(shape, biort, dtype=torch.float32):
    x = torch.randn(*shape, device=dev).to(dtype)
    sl = pw.ScatLayer(biort=biort).to(dev).to(dtype)
    rec = {'case': 'scat', 'shape': shape, 'biort': biort, 'dtype': str(dtype).split('.')[-1]}
    for ns in (0, 1):
        ops.set_option('no_stream', ns)
        try:
            with torch.no_grad():
                z = sl(x)
                c0 = pw.launch_count(); sl(x); kf = pw.kernels_since(c0)
                tf = min(bench.time_seq_fn(lambda: sl(x), 30, sync) for _ in range(3))
            xg = x.clone().requires_grad_(True)
            def step():
                zz = sl(xg)
                torch.autograd.grad(zz, xg, zz)
            c0 = pw.launch_count(); step(); kb = pw.kernels_since(c0)
            tb = min(bench.time_seq_fn(step, 20, sync) for _ in range(2))
        finally:
            ops.set_option('no_stream', 0)
        b = x.element_size() * (x.numel() + z.numel())
        tag = 'tile' if ns else 'stream'
        rec.update({tag + '_fwd_ms': round(tf, 4), tag + '_fwd_frac': round(b / tf / 8e9, 3), tag + '_fwdbwd_ms': round(tb, 4), tag + '_k': short(kf) + ' | ' + short(kb)})
    print(json.dumps(rec), flush=True)


dt((64, 3, 512, 512), 3, 'near_sym_a', 'qshift_a')
for J in (1, 2, 3):
    dt((64, 3, 512, 512), J, 'near_sym_b', 'qshift_b')
dt((64, 3, 512, 512), 3, 'near_sym_b', 'qshift_d')
dt((16, 3, 1024, 1024), 3, 'near_sym_b', 'qshift_b')
dt((256, 3, 256, 256), 2, 'near_sym_b', 'qshift_b')
dt((256, 3, 224, 224), 2, 'near_sym_b', 'qshift_b')
dt((512, 3, 128, 128), 1, 'near_sym_b', 'qshift_b')
dt((64, 3, 512, 512), 3, 'near_sym_b', 'qshift_b', torch.float16)
dt((64, 3, 512, 512), 3, 'legall', 'qshift_06')
dt((64, 3, 512, 512), 3, 'antonini', 'qshift_c')
scat((64, 3, 256, 256), 'near_sym_a')
scat((64, 3, 256, 256), 'near_sym_b')
scat((64, 3, 256, 256), 'near_sym_b_bp')
scat((256, 3, 256, 256), 'near_sym_b_bp')
scat((64, 3, 512, 512), 'near_sym_b_bp')
scat((16, 3, 1024, 1024), 'near_sym_b_bp')
scat((256, 3, 256, 256), 'near_sym_b_bp', torch.float16)
scat((256, 3, 256, 256), 'near_sym_a')
scat((256, 3, 256, 256), 'near_sym_b')
scat((64, 3, 512, 512), 'near_sym_b')
scat((128, 3, 224, 224), 'near_sym_b')
scat((1024, 3, 128, 128), 'near_sym_b')
scat((256, 3, 256, 256), 'near_sym_b', torch.float16)
