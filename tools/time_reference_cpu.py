"""Time the REAL reference (fbcotter/pytorch_wavelets from /root/reference, CPU) on BASELINE configs[1] in the authoring
container, next to oracle/torch_cpu.py on the same cores.   PYTHONPATH=tools/ref_shim:/root/reference python tools/time_reference_cpu.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets as ref
from oracle import torch_cpu as tc
from pytorch_wavelets_amd import filters
torch.manual_seed(0)
x = torch.randn(16, 3, 512, 512)
xfm, ifm = ref.DWTForward(J=3, wave='db4', mode='symmetric'), ref.DWTInverse(wave='db4', mode='symmetric')
h0, h1 = filters.dwt_analysis_taps('db4'); g0, g1 = filters.dwt_synthesis_taps('db4')


def timed(fn):
    fn()
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 8:
        fn(); reps += 1
    return x.numel() / ((time.perf_counter() - t0) / reps) / 1e6


with torch.no_grad():
    r = timed(lambda: ifm(xfm(x)))
    t = timed(lambda: tc.dwt_inverse(*tc.dwt_forward(x, 3, h0, h1, 'symmetric'), g0, g1, 'symmetric'))
out = {'workload': 'DWTForward+DWTInverse J=3 db4 symmetric, 16x3x512x512 fp32, CPU', 'cpu_count': os.cpu_count(),
       'torch_threads': torch.get_num_threads(), 'cpu': open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t'),
       'reference_mpix_s': round(r, 2), 'restated_torch_cpu_mpix_s': round(t, 2), 'torch': torch.__version__}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r02_reference_cpu_timing.json'), 'w'), indent=1)
print(out)
