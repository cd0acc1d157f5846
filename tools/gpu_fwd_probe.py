"""Time DWTForward (db4, symmetric, 128x3x512x512 fp32) for J in argv with the fused streaming kernel
and with the per-level generic kernels (WL_DISABLE_FUSED=1), HIP-event timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
HH = int(os.environ.get('PROBE_H', 512)); WW = int(os.environ.get('PROBE_W', 512))
NN = int(os.environ.get("PROBE_N", 128))
x = torch.randn(NN, 3, HH, WW, device=dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
Js = [int(v) for v in sys.argv[1:]] or [1, 2, 3]
tag = ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('WL_'))
for J in Js:
    xfm = pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev)
    with torch.no_grad():
        a = t(lambda: xfm(x))
    nb = NN * 3 * 4 * (HH * WW + 4 * ((HH + 7) // 2) * ((WW + 7) // 2)) if J == 1 else 0
    print('%-30s N=%d %dx%d J=%d fwd %.4f ms (%.3f us/plane) %s' % (tag, NN, HH, WW, J, a, a * 1e3 / (3 * NN), ('%.0f GB/s' % (nb / a / 1e6)) if nb else ''), flush=True)
