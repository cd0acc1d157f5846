"""GPU probe: wall-clock per step of DTCWT fwd+inv over time (does the host side keep up?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
xfm, ifm = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
x = torch.randn(64, 3, 512, 512, device=dev)
with torch.no_grad():
    for blk in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25):
            c = xfm(x)
            r = ifm(c)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print('block %d: host %.3f ms/step, with drain %.3f ms/step; reserved %.1f GB, alloc retries %d, segments %d' % (
            blk, (t1 - t0) / 25 * 1e3, (t2 - t0) / 25 * 1e3, st['reserved_bytes.all.current'] / 1e9, st['num_alloc_retries'], st['segment.all.current']), flush=True)
    # forward only / inverse only
    for name, fn in (('fwd', lambda: xfm(x)), ('inv', lambda: ifm(c))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(name, 'host %.3f ms, drained %.3f ms' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
