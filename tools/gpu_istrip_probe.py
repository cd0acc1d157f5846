"""GPU probe: the one-level streaming strip SYNTHESIS kernel (wl_dwt2d_synthesis_stream) against the per-level tile kernel:
values (max relative difference) and time per launch."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops, filters
from pytorch_wavelets_amd.dwt import lowlevel as ll

dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = [('cfg5 L1', 'db8', 'periodization', (32, 16, 1024, 1024), torch.float16),
         ('cfg5 L2', 'db8', 'periodization', (32, 16, 512, 512), torch.float16),
         ('cfg5 L3', 'db8', 'periodization', (32, 16, 256, 256), torch.float16),
         ('cfg5 L4', 'db8', 'periodization', (32, 16, 128, 128), torch.float16),
         ('1024 db4 per', 'db4', 'periodization', (16, 3, 512, 512), torch.float32),
         ('512 db4 per', 'db4', 'periodization', (128, 3, 256, 256), torch.float32),
         ('1024 db8 sym', 'db8', 'symmetric', (16, 3, 520, 520), torch.float32),
         ('2048 db2 zero fp16', 'db2', 'zero', (8, 3, 1032, 1032), torch.float16),
         ('512 db8 sym', 'db8', 'symmetric', (128, 3, 263, 263), torch.float32), ('512 db4 sym', 'db4', 'symmetric', (128, 3, 259, 259), torch.float32),
         ('1024 db4 sym', 'db4', 'symmetric', (16, 3, 515, 515), torch.float32), ('1024 db4 sym L2', 'db4', 'symmetric', (16, 3, 261, 261), torch.float32)]
if os.environ.get('PROBE') == 'short':
    cases = cases[:2]
if os.environ.get('PROBE') == 'odd':
    cases = cases[-4:]
for tag, wave, mode, cshape, dt in cases:
    g0, g1 = filters.dwt_synthesis_taps(wave)
    tg = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (g0, g1, g0, g1)]
    N, C, Kh, Kw = cshape
    l = torch.randn(N, C, Kh, Kw, device=dev, dtype=dt)
    h = torch.randn(N, C, 3, Kh, Kw, device=dev, dtype=dt)
    m = ll.mode_to_int(mode)
    ref = ops.sfb2d(l, h, *tg, m)
    kt = pw.last_kernel()
    res = ops.sfb2d_stream(l, h, *tg, m, force=True)
    out = {'lib': os.environ.get('WL_LIB', ''), 'case': tag, 'coeffs': list(cshape), 'dtype': str(dt), 'tile_kernel': kt}
    if res is None:
        out['stream'] = 'declined'
    else:
        out['stream_kernel'] = pw.last_kernel()
        out['max_rel_diff'] = float((res.float() - ref.float()).abs().max() / ref.float().abs().max())
        t_tile = timeit(lambda: ops.sfb2d(l, h, *tg, m))
        t_str = timeit(lambda: ops.sfb2d_stream(l, h, *tg, m, force=True))
        b = (4 * l.numel() + ref.numel()) * l.element_size()
        out.update(tile_ms=round(t_tile, 4), stream_ms=round(t_str, 4), tile_frac=round(b / t_tile / 8e9, 4),
                   stream_frac=round(b / t_str / 8e9, 4))
    print(json.dumps(out), flush=True)
    del l, h, ref, res
