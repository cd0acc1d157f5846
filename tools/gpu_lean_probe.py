"""GPU probe: lean level-1 / ScatLayer streaming kernels (wl_dtcwt_fused.h MODE 0 / 1) against the tile kernels (no_stream)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.get()


def timeit(fn, n=30):
    for _ in range(15):
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
    for tag, mk, shape, bpp in (('scatlayer 256x3x256x256', lambda: pw.ScatLayer().to(dev), (256, 3, 256, 256), 11),
                                ('scatlayer 64x3x512x512', lambda: pw.ScatLayer().to(dev), (64, 3, 512, 512), 11),
                                ('scatlayer 16x3x1024x1024', lambda: pw.ScatLayer().to(dev), (16, 3, 1024, 1024), 11),
                                ('dtcwt J=1 fwd 64x3x512x512', lambda: pw.DTCWTForward(J=1).to(dev), (64, 3, 512, 512), 20),
                                ('dtcwt J=1 fwd 256x3x256x256', lambda: pw.DTCWTForward(J=1).to(dev), (256, 3, 256, 256), 20),
                                ('dtcwt J=3 fwd 64x3x512x512', lambda: pw.DTCWTForward(J=3).to(dev), (64, 3, 512, 512), 20),
                                ('scatlayerj2 64x3x256x256', lambda: pw.ScatLayerj2().to(dev), (64, 3, 256, 256), 0)):
        m = mk()
        x = torch.randn(*shape, device=dev)
        out = {'case': tag}
        res = {}
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            y = m(x)
            k = pw.last_kernel()
            t = timeit(lambda: m(x))
            res[ns] = y
            out['tile' if ns else 'stream'] = {'ms': round(t, 4), 'frac': round(bpp * x.numel() / t / 8e9, 4), 'last_kernel': k}
        lib.wl_set_option(b'no_stream', 0)
        a, b = res[0], res[1]
        if isinstance(a, tuple):
            d = max(float((a[0] - b[0]).abs().max() / b[0].abs().max()), max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(a[1], b[1])))
        else:
            d = float((a - b).abs().max() / b.abs().max())
        out['max_rel_diff'] = d
        print(json.dumps(out), flush=True)
