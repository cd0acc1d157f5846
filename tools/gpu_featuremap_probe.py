"""Wavelet pooling inside a CNN: DWTForward / DWTInverse J = 1 on ImageNet-style feature maps (N = 64), forward, inverse and the
training step of the forward - which kernels run and what fraction of the HBM roofline they reach."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'; sync = torch.cuda.synchronize
for wave, L in (('haar', 2), ('db2', 4)):
    for shape in ((64, 64, 112, 112), (64, 128, 56, 56), (64, 256, 28, 28), (64, 512, 14, 14), (64, 64, 128, 128), (32, 64, 224, 224)):
        x = torch.randn(*shape, device=dev)
        fx, fi = pw.DWTForward(J=1, wave=wave, mode='zero').to(dev), pw.DWTInverse(wave=wave, mode='zero').to(dev)
        with torch.no_grad():
            c = fx(x)
            c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
            c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
            tf, ti = bench.time_seq_fn(lambda: fx(x), 20, sync), bench.time_seq_fn(lambda: fi(c), 20, sync)

        def step():
            xg = x.detach().requires_grad_(True)
            yl, yh = fx(xg)
            return torch.autograd.grad(yl.sum() + yh[0].sum(), xg)
        tt = bench.time_seq_fn(step, 10, sync)
        b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], 1, L, 4)
        print('%s %s: fwd %.4f ms %.3f %s  inv %.4f ms %.3f %s  fwd+bwd %.4f ms %.3f' % (
            wave, shape, tf, b / tf / 8e9, [k.split('<')[0] for k in kf], ti, b / ti / 8e9, [k.split('<')[0] for k in ki], tt, 2 * b / tt / 8e9), flush=True)
        del x, c
