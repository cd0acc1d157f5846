"""GPU probe: levels 1 + 2 of the DTCWT forward in one launch (wl_dtcwt_fused.h) against the per-level kernels
(wl_set_option no_stream)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.get()


def timeit(fn, n=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for J, shape, dt in ((2, (64, 3, 512, 512), torch.float32), (3, (64, 3, 512, 512), torch.float32),
                         (2, (16, 3, 1024, 1024), torch.float32), (2, (256, 3, 256, 256), torch.float32),
                         (2, (64, 3, 512, 512), torch.float16), (3, (8, 3, 512, 512), torch.float32)):
        m = pw.DTCWTForward(J=J).to(dev).to(dt)
        x = torch.randn(*shape, device=dev).to(dt)
        # algorithmic bytes per pixel of J levels: x + highs_j (12 / 4^j) + the last lowpass
        bpp = (1 + sum(12 / 4 ** j for j in range(1, J + 1)) + 1 / 4 ** (J - 1)) * x.element_size()
        out = {'case': 'dtcwt J=%d fwd %s %s' % (J, 'x'.join(map(str, shape)), str(dt).split('.')[-1]), 'bytes_per_px': round(bpp, 3)}
        res = {}
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            y = m(x)
            t = timeit(lambda: m(x))
            res[ns] = y
            out['per_level_tile' if ns else 'fused'] = {'ms': round(t, 4), 'frac': round(bpp * x.numel() / t / 8e9, 4)}
        lib.wl_set_option(b'no_stream', 0)
        a, b = res[0], res[1]
        out['max_rel_diff'] = max(float((a[0].float() - b[0].float()).abs().max() / b[0].float().abs().max()),
                                  max(float((p.float() - q.float()).abs().max() / q.float().abs().max()) for p, q in zip(a[1], b[1])))
        print(json.dumps(out), flush=True)
