"""GPU probe: the one-level streaming strip kernel (wl_dwt2d_analysis_stream) against the per-level tile kernel -
values (max relative difference) and time per launch, per level of the configurations outside the fused envelope."""
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


cases = [('cfg5 L1', 'db8', 'periodization', (32, 16, 2048, 2048), torch.float16),
         ('cfg5 L2', 'db8', 'periodization', (32, 16, 1024, 1024), torch.float16),
         ('cfg5 L3', 'db8', 'periodization', (32, 16, 512, 512), torch.float16),
         ('cfg5 L4', 'db8', 'periodization', (32, 16, 256, 256), torch.float16),
         ('1024 db4 L1', 'db4', 'symmetric', (16, 3, 1024, 1024), torch.float32),
         ('1024 db4 L2', 'db4', 'symmetric', (16, 3, 516, 516), torch.float32),
         ('512 db8 L1', 'db8', 'symmetric', (128, 3, 512, 512), torch.float32),
         ('512 db8 L2', 'db8', 'symmetric', (128, 3, 264, 264), torch.float32),
         ('512 db4 L1', 'db4', 'symmetric', (128, 3, 512, 512), torch.float32),
         ('2048 db4 fp32', 'db4', 'symmetric', (8, 3, 2048, 2048), torch.float32),
         ('512 db4 per', 'db4', 'periodization', (128, 3, 512, 512), torch.float32),
         ('4096 db2 fp16', 'db2', 'zero', (4, 3, 4096, 4096), torch.float16),
         ('1024 db4 L2 odd', 'db4', 'symmetric', (16, 3, 515, 515), torch.float32), ('512 db4 L2 odd', 'db4', 'symmetric', (128, 3, 259, 259), torch.float32),
         ('2048 db4 L2 odd', 'db4', 'symmetric', (8, 3, 1027, 1027), torch.float32)]
if os.environ.get('PROBE') == 'odd':
    cases = cases[-3:]
if os.environ.get('PROBE') == 'short':
    cases = [c for c in cases if c[0] in ('cfg5 L1', 'cfg5 L3', '512 db8 L1', '1024 db4 L1')]
for tag, wave, mode, shape, dt in cases:
    h0, h1 = filters.dwt_analysis_taps(wave)
    th = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (h0, h1, h0, h1)]
    x = torch.randn(*shape, device=dev, dtype=dt)
    m = ll.mode_to_int(mode)
    ref = ops.afb2d(x, *th, m)
    kt = pw.last_kernel()
    res = ops.afb2d_stream(x, *th, m, force=True)
    out = {'lib': os.environ.get('WL_LIB', ''), 'case': tag, 'shape': list(shape), 'dtype': str(dt), 'tile_kernel': kt}
    if res is None:
        out['stream'] = 'declined'
    else:
        out['stream_kernel'] = pw.last_kernel()
        out['max_rel_diff'] = max(float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(res, ref))
        t_tile = timeit(lambda: ops.afb2d(x, *th, m))
        t_str = timeit(lambda: ops.afb2d_stream(x, *th, m, force=True))
        L = len(h0)
        per = mode == 'periodization'
        kh = (shape[2] + 1) // 2 if per else (shape[2] + L - 1) // 2
        kw = (shape[3] + 1) // 2 if per else (shape[3] + L - 1) // 2
        b = shape[0] * shape[1] * (shape[2] * shape[3] + 4 * kh * kw) * x.element_size()
        out.update(tile_ms=round(t_tile, 4), stream_ms=round(t_str, 4), tile_frac=round(b / t_tile / 8e9, 4),
                   stream_frac=round(b / t_str / 8e9, 4), level_bytes=b)
    print(json.dumps(out), flush=True)
    del x, ref, res
