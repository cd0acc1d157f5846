"""GPU probe: the streaming level-1 DTCWT inverse strip kernel against the tile kernel (wl_set_option no_stream)."""
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
    for tag, J, shape in (('dtcwt J=1 inv 64x3x512x512', 1, (64, 3, 512, 512)), ('dtcwt J=3 inv 64x3x512x512', 3, (64, 3, 512, 512)),
                          ('dtcwt J=1 inv 16x3x1024x1024', 1, (16, 3, 1024, 1024)), ('dtcwt J=1 inv 256x3x256x256', 1, (256, 3, 256, 256))):
        xfm, ifm = pw.DTCWTForward(J=J).to(dev), pw.DTCWTInverse().to(dev)
        x = torch.randn(*shape, device=dev)
        c = xfm(x)
        out = {'case': tag}
        res = {}
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            res[ns] = ifm(c)
            k = pw.last_kernel()
            t = timeit(lambda: ifm(c))
            out['tile' if ns else 'stream'] = {'ms': round(t, 4), 'frac': round(20 * x.numel() / t / 8e9, 4), 'last_kernel': k}
        lib.wl_set_option(b'no_stream', 0)
        out['max_rel_diff'] = float((res[0] - res[1]).abs().max() / res[1].abs().max())
        out['roundtrip_err'] = float((res[0] - x).abs().max())
        print(json.dumps(out), flush=True)
