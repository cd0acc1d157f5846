"""Time DWTInverse (db4, symmetric, 128x3x512x512 fp32) for J in argv, HIP-event timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
x = torch.randn(128, 3, 512, 512, device=dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('WL_'))
for J in [int(v) for v in sys.argv[1:]] or [1, 3]:
    xfm = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev)
    ifm = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
    with torch.no_grad():
        yl, yh = xfm(x)
        a = t(lambda: ifm((yl, yh)))
        err = float((ifm((yl, yh)) - x).abs().max())
    nb = 128 * 3 * 4 * (512 * 512 + 4 * 259 * 259) if J == 1 else 0
    print('%-30s J=%d inv %.4f ms %s  roundtrip err %.2e' % (tag, J, a, ('%.0f GB/s' % (nb / a / 1e6)) if nb else '', err), flush=True)
