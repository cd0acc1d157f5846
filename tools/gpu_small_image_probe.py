"""Small images, large batches (CIFAR / Tiny-ImageNet shapes, where the scattering layers are used upstream): which kernels run and
what fraction of the HBM roofline they reach.  usage: python tools/gpu_small_image_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pytorch_wavelets_amd as pw

dev = 'cuda:0'
sync = torch.cuda.synchronize
for shape in ((2048, 3, 32, 32), (1024, 3, 64, 64), (256, 3, 128, 128), (512, 16, 32, 32)):
    x = torch.randn(*shape, device=dev)
    P = x.numel()
    with torch.no_grad():
        sl = pw.ScatLayer().to(dev)
        c0 = pw.launch_count(); sl(x); k = pw.kernels_since(c0)
        t = bench.time_seq_fn(lambda: sl(x), 30, sync)
        print('%s ScatLayer: %.4f ms = %.3f of 8 TB/s at 11 B/px %s' % (shape, t, 11 * P / t / 8e9, k), flush=True)

        def step():
            xg = x.detach().requires_grad_(True)
            return torch.autograd.grad(sl(xg).sum(), xg)
        with torch.enable_grad():
            c0 = pw.launch_count(); step(); k = pw.kernels_since(c0)
            t = bench.time_seq_fn(step, 20, sync)
        print('%s ScatLayer fwd+bwd: %.4f ms = %.3f at 46 B/px %s' % (shape, t, 46 * P / t / 8e9, k), flush=True)
        for J in (1, 2):
            fx, fi = pw.DWTForward(J=J, wave='db2', mode='symmetric').to(dev), pw.DWTInverse(wave='db2', mode='symmetric').to(dev)
            c = fx(x)
            c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
            c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
            tf, ti = bench.time_seq_fn(lambda: fx(x), 30, sync), bench.time_seq_fn(lambda: fi(c), 30, sync)
            b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], J, 4, 4)
            print('%s DWT db2 J=%d: fwd %.4f ms = %.3f %s  inv %.4f ms = %.3f %s' % (shape, J, tf, b / tf / 8e9, kf, ti, b / ti / 8e9, ki), flush=True)
        dx, di = pw.DTCWTForward(J=2).to(dev), pw.DTCWTInverse().to(dev)
        yl, yh = dx(x)
        c0 = pw.launch_count(); dx(x); kf = pw.kernels_since(c0)
        c0 = pw.launch_count(); di((yl, yh)); ki = pw.kernels_since(c0)
        tf, ti = bench.time_seq_fn(lambda: dx(x), 30, sync), bench.time_seq_fn(lambda: di((yl, yh)), 30, sync)
        b = 4 * (P + yl.numel() + sum(h.numel() for h in yh))
        print('%s DTCWT J=2: fwd %.4f ms = %.3f %s  inv %.4f ms = %.3f %s' % (shape, tf, b / tf / 8e9, kf, ti, b / ti / 8e9, ki), flush=True)
