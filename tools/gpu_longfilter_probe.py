"""Long filters on the per-level tile kernels (fp32 db8 / db10, fp16 db10): forward / inverse time per level-1 launch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(f, n=20):
    for _ in range(10): f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


res = {'lib': os.environ.get('WL_LIB')}
with torch.no_grad():
    for dt, name in ((torch.float32, 'f32'), (torch.float16, 'f16')):
        x = torch.randn(64, 3, 512, 512, device=dev).to(dt)
        for wave in ('db4', 'db8', 'db10'):
            for mode in ('symmetric', 'periodization'):
                fx = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev).to(dt)
                ix = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
                yl, yh = fx(x)
                res['%s_%s_%s' % (name, wave, mode[:3])] = [timed(lambda: fx(x)), timed(lambda: ix((yl, yh)))]
print(json.dumps(res))
