"""Why was the db6 (12 taps) forward 7x slow when timed after another wavelet's runs?  Per-call wall times with a synchronize each,
and the allocator's device-malloc counter around them."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.dwt import lowlevel as _ll

dev = 'cuda:0'
x = torch.randn(128, 3, 512, 512, device=dev)


def stats():
    s = torch.cuda.memory_stats()
    return s.get('num_device_alloc', -1), s.get('num_device_free', -1), s.get('num_alloc_retries', -1), torch.cuda.memory_reserved() >> 20


for wave in ('db5', 'db6'):
    fx, fi = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev), pw.DWTInverse(wave=wave, mode='symmetric').to(dev)
    with torch.no_grad():
        c = fx(x)
        for fused in (True, False):
            _ll.FUSED_LEVELS = fused
            for name, fn in (('inv', lambda: fi(c)), ('fwd', lambda: fx(x))):
                s0 = stats()
                t = bench.time_seq_fn(fn, 20, torch.cuda.synchronize)
                s1 = stats()
                walls = []
                for _ in range(6):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e3)
                print('%s fused=%d %s: events %.4f ms; device mallocs %d -> %d, frees %d -> %d, retries %d, reserved %d MiB; wall per call %s' % (
                    wave, fused, name, t, s0[0], s1[0], s0[1], s1[1], s1[2], s1[3], ' '.join('%.2f' % w for w in walls)), flush=True)
        _ll.FUSED_LEVELS = True
