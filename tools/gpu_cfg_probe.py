"""Time the other BASELINE configs on the GPU: DTCWT fwd+inv (configs[2]), ScatLayer (configs[3] per-GPU share and
full), DWT J=4 db8 periodization fp16 (configs[4], reduced N)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    x = torch.randn(64, 3, 512, 512, device=dev)
    xfm, ifm = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
    yl, yh = xfm(x)
    tf, ti = t(lambda: xfm(x)), t(lambda: ifm((yl, yh)))
    px = x.numel()
    print('DTCWT J=3 64x3x512x512: fwd %.3f ms (%.0f GB/s alg) inv %.3f ms  fwd+inv %.0f Mpix/s' % (tf, 20 * px / tf / 1e6, ti, px / (tf + ti) / 1e3), flush=True)
    for j in (1, 2):
        xj = pw.DTCWTForward(J=j).to(dev)
        print('   DTCWT J=%d fwd %.3f ms' % (j, t(lambda: xj(x))), flush=True)
    del yl, yh
    for n in (32, 256):
        xs = torch.randn(n, 3, 256, 256, device=dev)
        sl = pw.ScatLayer().to(dev)
        ts = t(lambda: sl(xs))
        print('ScatLayer %dx3x256x256: %.3f ms  %.0f Mpix/s  (%.0f GB/s alg)' % (n, ts, xs.numel() / ts / 1e3, 11 * xs.numel() / ts / 1e6), flush=True)
    xh = torch.randn(8, 16, 2048, 2048, device=dev).half()
    x4 = pw.DWTForward(J=4, wave='db8', mode='periodization').to(dev).half()
    th = t(lambda: x4(xh), 5)
    print('DWT J=4 db8 per fp16 8x16x2048x2048: %.3f ms  %.0f Mpix/s (%.0f GB/s alg)' % (th, xh.numel() / th / 1e3, 4 * xh.numel() / th / 1e6), flush=True)
