"""Round 5: DTCWT J=3 / J=2 forward and inverse at config 3's shape, this package or (WL_PKG_ROOT=ab/old_pkg) round 4's, one fresh process
per run - the same-box check that the fused forward's reversed column taps (exact for any level-1 taps) cost nothing."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('WL_PKG_ROOT'):
    sys.path.insert(0, os.path.join(ROOT, os.environ['WL_PKG_ROOT']))
import pytorch_wavelets_amd as pw
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
out = {'pkg': os.environ.get('WL_PKG_ROOT', 'new')}
def t(name, fn, n=40):
    with torch.no_grad():
        out[name] = round(min(bench.time_seq_fn(fn, n, sync) for _ in range(5)), 4)
xd = torch.randn(64, 3, 512, 512, device=dev)
d = pw.DTCWTForward(J=3).to(dev); t('fwd_j3', lambda: d(xd))
yl, yh = d(xd); di = pw.DTCWTInverse().to(dev); t('inv_j3', lambda: di((yl, yh)))
d2 = pw.DTCWTForward(J=2).to(dev); t('fwd_j2', lambda: d2(xd))
d1 = pw.DTCWTForward(J=1).to(dev); t('fwd_j1', lambda: d1(xd))
s = pw.ScatLayer().to(dev); xs = torch.randn(256, 3, 256, 256, device=dev); t('scat', lambda: s(xs))
xg = xs.clone().requires_grad_(True)
def train():
    with torch.enable_grad():
        z = s(xg); z.backward(torch.ones_like(z))
out['scat_train'] = round(min(bench.time_seq_fn(train, 20, sync) for _ in range(3)), 4)
del xg
x2 = torch.randn(64, 3, 256, 256, device=dev)
s2 = pw.ScatLayerj2().to(dev); t('scatj2', lambda: s2(x2))
sr = pw.ScatLayer(biort='near_sym_b_bp').to(dev); t('scat_rot', lambda: sr(x2))
x5 = torch.randn(64, 3, 512, 512, device=dev); t('scat_512', lambda: s(x5))
print(json.dumps(out), flush=True)
