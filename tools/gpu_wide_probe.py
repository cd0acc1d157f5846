"""DWT J=3 forward / inverse outside the metric's shape: wider planes, longer filters, larger batches - which kernels run and
what fraction of the HBM roofline (algorithmic bytes) they reach.  usage: python tools/gpu_wide_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pytorch_wavelets_amd as pw

dev = 'cuda:0'
for shape, wave, L in (((16, 3, 1024, 1024), 'db4', 8), ((64, 3, 1024, 1024), 'db4', 8), ((128, 3, 512, 512), 'db8', 16),
                       ((32, 3, 2048, 2048), 'db4', 8), ((128, 3, 512, 512), 'db6', 12), ((256, 3, 256, 256), 'db4', 8)):
    x = torch.randn(*shape, device=dev)
    fx, fi = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev), pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
    with torch.no_grad():
        c = fx(x)
        c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
        c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
        tf = bench.time_seq_fn(lambda: fx(x), 30, torch.cuda.synchronize)
        ti = bench.time_seq_fn(lambda: fi(c), 30, torch.cuda.synchronize)
    b = bench.algorithmic_bytes_fwd(shape[0], shape[1], shape[2], shape[3], 3, L, 4)
    print('%s %s: fwd %.4f ms = %.3f %s   inv %.4f ms = %.3f %s' % (shape, wave, tf, b / tf / 1e6 / 8000, kf, ti, b / ti / 1e6 / 8000, ki), flush=True)
    del x, c
