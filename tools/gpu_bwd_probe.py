"""Forward+backward timings (autograd through the engine): ScatLayer, DWT, DTCWT."""
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
def fb(mod, x, red):
    x.grad = None
    out = mod(x)
    red(out).backward()
x = torch.randn(256, 3, 256, 256, device=dev, requires_grad=True)
sl = pw.ScatLayer().to(dev)
with torch.no_grad():
    tf = t(lambda: sl(x.detach()))
tfb = t(lambda: fb(sl, x, lambda z: z.sum()))
print('ScatLayer 256x3x256x256: fwd(no grad) %.3f ms  fwd+bwd (incl. sum + grad of sum) %.3f ms' % (tf, tfb), flush=True)
x = torch.randn(128, 3, 512, 512, device=dev, requires_grad=True)
m = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
with torch.no_grad():
    tf = t(lambda: m(x.detach()))
tfb = t(lambda: fb(m, x, lambda o: o[0].sum() + sum(h.sum() for h in o[1])))
print('DWT J=3 128x3x512x512: fwd %.3f ms  fwd+bwd %.3f ms' % (tf, tfb), flush=True)
x = torch.randn(64, 3, 512, 512, device=dev, requires_grad=True)
m = pw.DTCWTForward(J=3).to(dev)
with torch.no_grad():
    tf = t(lambda: m(x.detach()))
tfb = t(lambda: fb(m, x, lambda o: o[0].sum() + sum(h.sum() for h in o[1])))
print('DTCWT J=3 64x3x512x512: fwd %.3f ms  fwd+bwd %.3f ms' % (tf, tfb), flush=True)
