"""Round 5: how many compute waves should a one-level DWT strip workgroup have on NARROW levels?  The strip kernels' workgroups are 4
compute + 4 stager waves cut for strips of 512 output columns (analysis) / 1024 (synthesis); the deeper levels of a wide pyramid
(config 5: 2048^2 float16 db8 periodization -> levels of 512 and 256 columns) leave half / three quarters of the compute waves idle.
WL_LIB selects an A/B build with fewer compute waves and more workgroups per CU (tools/build_ab_strip.sh cw2 / cw1); one fresh process
per run."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
out = {'lib': os.environ.get('WL_LIB', 'product')}
ops.STREAM_FORCE = True
def t(name, fn, n=40):
    with torch.no_grad():
        out[name] = round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
for dt, wave, mode, tag in ((torch.float16, 'db8', 'periodization', 'h16per'), (torch.float32, 'db4', 'symmetric', 'f32sym'), (torch.float32, 'db8', 'symmetric', 'f32db8')):
    f = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev).to(dt)
    i = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
    for W in (1024, 512, 256):
        planes = 512 * (512 * 512) // (W * W) if W < 1024 else 256
        x = torch.randn(planes // 16, 16, W, W, device=dev).to(dt)
        with torch.no_grad():
            c0 = pw.launch_count()
            yl, yh = f(x)
            i((yl, yh))
            kern = pw.kernels_since(c0)
        t('%s_fwd_%d' % (tag, W), lambda: f(x))
        t('%s_inv_%d' % (tag, W), lambda: i((yl, yh)))
        if W == 512:
            out['%s_kern_%d' % (tag, W)] = kern
        del x, yl, yh
print(json.dumps(out), flush=True)
