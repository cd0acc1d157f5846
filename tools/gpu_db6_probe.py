"""db6 (12 taps) J=3 on 128x3x512x512: forward and inverse timed alternately (round 4: whichever ran second was 7x slow)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pytorch_wavelets_amd as pw

dev = 'cuda:0'
x = torch.randn(128, 3, 512, 512, device=dev)
wave = sys.argv[1] if len(sys.argv) > 1 else 'db6'
fx, fi = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev), pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
with torch.no_grad():
    c = fx(x)
    for rep in range(3):
        tf = bench.time_seq_fn(lambda: fx(x), 20, torch.cuda.synchronize)
        ti = bench.time_seq_fn(lambda: fi(c), 20, torch.cuda.synchronize)
        print('rep %d: fwd %.4f ms  inv %.4f ms' % (rep, tf, ti), flush=True)
    for rep in range(2):
        ti = bench.time_seq_fn(lambda: fi(c), 20, torch.cuda.synchronize)
        tf = bench.time_seq_fn(lambda: fx(x), 20, torch.cuda.synchronize)
        print('rep %d: inv %.4f ms  fwd %.4f ms' % (rep, ti, tf), flush=True)
    c0 = pw.launch_count(); fx(x); print(pw.kernels_since(c0)); c0 = pw.launch_count(); fi(c); print(pw.kernels_since(c0))
