"""Round 5: DTCWT J=3 forward / inverse at config 3's shape after a clock ramp (300 untimed calls), this package or round 4's (WL_PKG_ROOT=ab/old_pkg)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('WL_PKG_ROOT'):
    sys.path.insert(0, os.path.join(ROOT, os.environ['WL_PKG_ROOT']))
import pytorch_wavelets_amd as pw
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
x = torch.randn(64, 3, 512, 512, device=dev)
d = pw.DTCWTForward(J=3).to(dev); di = pw.DTCWTInverse().to(dev)
with torch.no_grad():
    c = d(x)
    for _ in range(300):
        d(x)
    f = [round(bench.time_seq_fn(lambda: d(x), 100, sync), 4) for _ in range(5)]
    for _ in range(300):
        di(c)
    i = [round(bench.time_seq_fn(lambda: di(c), 100, sync), 4) for _ in range(5)]
print(json.dumps({'pkg': os.environ.get('WL_PKG_ROOT', 'round5'), 'fwd': f, 'inv': i}), flush=True)
