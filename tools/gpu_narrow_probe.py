"""GPU probe: the one-level strip kernels on NARROW levels (rows of 512 B - 1 KiB), forced, against the tile kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops, filters
dev = torch.device('cuda:0')


def timeit(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[1]


with torch.no_grad():
    for tag, shape, wave, mode, dt in (('cfg5 L3', (512, 512, 512), 'db8', 2, torch.float16), ('cfg5 L4', (512, 256, 256), 'db8', 2, torch.float16),
                                       ('db8 512^2 L2', (384, 256, 256), 'db8', 1, torch.float32), ('db8 512^2 L3', (384, 128, 128), 'db8', 1, torch.float32),
                                       ('db4 1024^2 L3', (48, 256, 256), 'db4', 1, torch.float32), ('db4 512 J=1 few planes', (48, 512, 512), 'db4', 1, torch.float32)):
        P, H, W = shape
        x = torch.randn(1, P, H, W, device=dev).to(dt)
        h0, h1 = filters.dwt_analysis_taps(wave)
        g0, g1 = filters.dwt_synthesis_taps(wave)
        th = [torch.tensor(v[::-1].copy(), dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
        tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
        out = {'case': tag, 'shape': shape, 'dtype': str(dt).split('.')[-1]}
        r_s = ops.afb2d_stream(x, *th, mode, force=True)
        if r_s is not None:
            out['fwd_strip_ms'] = round(timeit(lambda: ops.afb2d_stream(x, *th, mode, force=True)), 4)
        r_t = ops.afb2d(x, *th, mode)
        out['fwd_tile_ms'] = round(timeit(lambda: ops.afb2d(x, *th, mode)), 4)
        if r_s is not None:
            out['fwd_diff'] = float((r_s[0].float() - r_t[0].float()).abs().max() / r_t[0].float().abs().max())
        ll, hs = r_t
        y_s = ops.sfb2d_stream(ll, hs, *tg, mode, force=True)
        if y_s is not None:
            out['inv_strip_ms'] = round(timeit(lambda: ops.sfb2d_stream(ll, hs, *tg, mode, force=True)), 4)
        y_t = ops.sfb2d(ll, hs, *tg, mode)
        out['inv_tile_ms'] = round(timeit(lambda: ops.sfb2d(ll, hs, *tg, mode)), 4)
        if y_s is not None:
            out['inv_diff'] = float((y_s.float() - y_t.float()).abs().max() / y_t.float().abs().max())
        print(json.dumps(out), flush=True)
