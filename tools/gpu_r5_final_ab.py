"""Round 5, last same-box A/B: the package of round 4 (WL_PKG_ROOT=ab/old_pkg: python + library of commit e0907db) against the final one,
through the public module API only, on the shapes this round worked on.  One JSON line; run the two alternately (tools/gpu_r5_final_ab.sh)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('WL_PKG_ROOT'):
    sys.path.insert(0, os.path.join(ROOT, os.environ['WL_PKG_ROOT']))
import pytorch_wavelets_amd as pw        # (before bench: importing bench puts the repo root in front of sys.path)
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
out = {'pkg': os.environ.get('WL_PKG_ROOT', 'round5')}
def t(name, fn, n=40):
    with torch.no_grad():
        fn(); fn()
        out[name] = round(min(bench.time_seq_fn(fn, n, sync) for _ in range(4)), 4)
def dwt(tag, shape, J, wave, mode, dt=torch.float32):
    x = torch.randn(*shape, device=dev).to(dt)
    f = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev).to(dt); i = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
    with torch.no_grad():
        c = f(x)
    t(tag + '_fwd', lambda: f(x)); t(tag + '_inv', lambda: i(c))
dwt('metric_128x3x512_db4_J3', (128, 3, 512, 512), 3, 'db4', 'symmetric')
dwt('db8_128x3x512_J3', (128, 3, 512, 512), 3, 'db8', 'symmetric')
dwt('per_128x3x512_db4_J3', (128, 3, 512, 512), 3, 'db4', 'periodization')
dwt('cfg5_32x16x2048_db8_J4_f16', (32, 16, 2048, 2048), 4, 'db8', 'periodization', torch.float16)
dwt('imnet_128x3x224_J1', (128, 3, 224, 224), 1, 'db4', 'symmetric')
dwt('imnet_128x3x224_J3', (128, 3, 224, 224), 3, 'db4', 'symmetric')
dwt('imnet_512x3x224_J3', (512, 3, 224, 224), 3, 'db4', 'symmetric')
dwt('imnet_64x3x224_J2', (64, 3, 224, 224), 2, 'db4', 'symmetric')
dwt('128x3x640_J3', (128, 3, 640, 640), 3, 'db4', 'symmetric')
dwt('64x3x1024_J3', (64, 3, 1024, 1024), 3, 'db4', 'symmetric')
dwt('32x3x2048_J3', (32, 3, 2048, 2048), 3, 'db4', 'symmetric')
dwt('512x3x128_J3', (512, 3, 128, 128), 3, 'db4', 'symmetric')
x = torch.randn(64, 3, 512, 512, device=dev)
d = pw.DTCWTForward(J=3).to(dev); di = pw.DTCWTInverse().to(dev)
with torch.no_grad():
    c = d(x)
t('dtcwt_64x3x512_J3_fwd', lambda: d(x)); t('dtcwt_64x3x512_J3_inv', lambda: di(c))
xs = torch.randn(256, 3, 256, 256, device=dev); s = pw.ScatLayer().to(dev); t('scat_256x3x256', lambda: s(xs))
xs = torch.randn(128, 3, 224, 224, device=dev); t('scat_128x3x224', lambda: s(xs))
print(json.dumps(out), flush=True)
