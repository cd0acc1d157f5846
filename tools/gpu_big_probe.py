"""GPU probe: very wide planes (4096 columns: many strips per plane) through the streaming kernels against the tile kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib, ops
dev = torch.device('cuda:0')
lib = _lib.get()
def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
for shape, dt in (((8, 1, 4096, 4096), torch.float32), ((4, 2, 2048, 4100), torch.float32), ((8, 1, 4096, 4096), torch.float16)):
    x = torch.randn(*shape, device=dev).to(dt)
    out = {'shape': shape, 'dtype': str(dt).split('.')[-1]}
    xfm, ifm, sl = pw.DTCWTForward(J=3).to(dev).to(dt), pw.DTCWTInverse().to(dev).to(dt), pw.ScatLayer().to(dev).to(dt)
    res = {}
    for ns in (0, 1):
        lib.wl_set_option(b'no_stream', ns)
        c0 = pw.launch_count()
        yl, yh = xfm(x); k1 = pw.kernels_since(c0)
        c0 = pw.launch_count()
        r = ifm((yl, yh)); k2 = pw.kernels_since(c0)
        z = sl(x)
        res[ns] = [yl] + list(yh) + [r, z]
        if ns == 0:
            out['kernels'] = k1 + k2 + [pw.last_kernel()]
    lib.wl_set_option(b'no_stream', 0)
    out['dtcwt_stream_vs_tile'] = max(rel(a, b) for a, b in zip(res[0], res[1]))
    out['dtcwt_roundtrip'] = rel(res[0][-2], x)
    del res
    with torch.no_grad():
        for wave, mode in (('db4', 'symmetric'), ('db8', 'periodization')):
            fx, fi = pw.DWTForward(J=3, wave=wave, mode=mode).to(dev).to(dt), pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
            c0 = pw.launch_count()
            yl, yh = fx(x)
            ks = pw.kernels_since(c0)
            r = fi((yl, yh))
            ops.FUSED_STRIPS = False if hasattr(ops, 'FUSED_STRIPS') else None
            out['dwt_%s_%s' % (wave, mode)] = {'roundtrip': rel(r, x), 'fwd_kernels': ks, 'last_inv': pw.last_kernel()}
    print(json.dumps(out), flush=True)
