"""Round-5 same-box A/B: the package of the previous round (WL_PKG_ROOT=ab/old_pkg: python + library) against this one on the launches the
tap-relation guards touch - 12-tap fused forward (one-bank variant + armed fallback), 16-tap strip kernels (QMF variant + armed
fallback: config 5 reduced to 8 planes... full size), the fused DTCWT forward (reversed column taps above / below the plane).
usage: python tools/gpu_r5_ab.py [tag]   (prints one JSON line per case)"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('WL_PKG_ROOT'):      # the package (python + library) of another commit, e.g. ab/old_pkg (git archive <commit> pytorch_wavelets_amd)
    sys.path.insert(0, os.path.join(ROOT, os.environ['WL_PKG_ROOT']))
import pytorch_wavelets_amd as pw        # (before bench: importing bench puts the repo root in front of sys.path)
from pytorch_wavelets_amd import ops
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get('WL_PKG_ROOT', 'new')
if os.environ.get('WL_NO_LATTICE'):
    ops.STRIP_LATTICE = False
    if hasattr(ops, 'ROWS_LATTICE'):
        ops.ROWS_LATTICE = False
if os.environ.get('WL_ROWS_LAT8'):          # with WL_LIB=ab/libwl_lat8.so (-DWL_ROWS_LAT_MIN=8): the lattice on the metric's forward kernel
    ops.ROWS_LATTICE_MIN = 8
dev = 'cuda:0'; sync = torch.cuda.synchronize
out = {}
def t(name, fn, n=30):
    with torch.no_grad():
        fn(); fn(); c0 = pw.launch_count(); fn(); ks = pw.kernels_since(c0)
        ms = min(bench.time_seq_fn(fn, n, sync) for _ in range(3))
    out[name] = round(ms, 4); out[name + '_k'] = [k.replace('float', 'f') for k in ks]
x = torch.randn(128, 3, 512, 512, device=dev)
for wave in ('db4', 'db5', 'db6'):
    m = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev); t('fwd_' + wave, lambda: m(x))
    yl, yh = m(x); im = pw.DWTInverse(wave=wave, mode='symmetric').to(dev); t('inv_' + wave, lambda: im((yl, yh)))
for wave in ('db7', 'db10', 'sym8'):
    mm = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev); t('fwd_' + wave, lambda: mm(x))
    yy = mm(x); ii = pw.DWTInverse(wave=wave, mode='symmetric').to(dev); t('inv_' + wave, lambda: ii(yy))
    del yy
m8 = pw.DWTForward(J=3, wave='db8', mode='symmetric').to(dev); t('fwd_db8', lambda: m8(x))
yl, yh = m8(x); i8 = pw.DWTInverse(wave='db8', mode='symmetric').to(dev); t('inv_db8', lambda: i8((yl, yh)))
del x, yl, yh
xd = torch.randn(64, 3, 512, 512, device=dev)
d = pw.DTCWTForward(J=3).to(dev); t('dtcwt_fwd', lambda: d(xd))
yl, yh = d(xd); di = pw.DTCWTInverse().to(dev); t('dtcwt_inv', lambda: di((yl, yh)))
d2 = pw.DTCWTForward(J=2).to(dev); t('dtcwt_fwd_j2', lambda: d2(xd))
del xd, yl, yh
xh = torch.randn(32, 16, 2048, 2048, device=dev, dtype=torch.float16)
m5 = pw.DWTForward(J=4, wave='db8', mode='periodization').to(dev).half(); t('cfg5_fwd', lambda: m5(xh), 5)
m51 = pw.DWTForward(J=1, wave='db8', mode='periodization').to(dev).half(); t('cfg5_fwd_l1', lambda: m51(xh), 5)
yl, yh = m5(xh); del xh
i5 = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half(); t('cfg5_inv', lambda: i5((yl, yh)), 5)
print(json.dumps({'lib': tag, **out}), flush=True)
