"""BASELINE configs[4] (DWT J=4 db8 periodization fp16, reduced batch) and its fp32 twin, per level count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tag = os.environ.get('WL_LIB', '')
with torch.no_grad():
    for dt in (torch.float16, torch.float32):
        xh = torch.randn(8, 16, 2048, 2048, device=dev).to(dt)
        for J in (1, 4):
            m = pw.DWTForward(J=J, wave='db8', mode='periodization').to(dev).to(dt)
            im = pw.DWTInverse(wave='db8', mode='periodization').to(dev).to(dt)
            th = t(lambda: m(xh))
            yl, yh = m(xh)
            ti = t(lambda: im((yl, yh)))
            b = 2 * xh.numel() * xh.element_size()
            print('%s %s J=%d fwd %.3f ms (%.0f GB/s alg, %.0f Mpix/s)  inv %.3f ms (%.0f GB/s)' % (tag, str(dt)[6:], J, th, b / th / 1e6, xh.numel() / th / 1e3, ti, b / ti / 1e6), flush=True)
