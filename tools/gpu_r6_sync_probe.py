import os, sys, time, ctypes
mode = sys.argv[1]
if mode == 'env':
    os.environ['ROC_ACTIVE_WAIT_TIMEOUT'] = '1000000'
import torch
if mode == 'flag':
    hip = ctypes.CDLL('libamdhip64.so')
    print('hipSetDeviceFlags rc', hip.hipSetDeviceFlags(1))
x = torch.randn(128*3, 512, 512, device='cuda')
y = torch.empty_like(x)
torch.cuda.synchronize()
for trial in range(3):
    lat = []
    for _ in range(50):
        for _ in range(3): y.copy_(x)
        e = torch.cuda.Event(enable_timing=False)
        t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        # now device idle: time an empty sync and a launch+sync
        t2 = time.perf_counter(); y[:1].copy_(x[:1]); torch.cuda.synchronize(); t3 = time.perf_counter()
        lat.append((t3 - t2) * 1e6)
    lat.sort()
    print(mode, 'tiny launch + sync on idle device: median %.1f us, min %.1f' % (lat[len(lat)//2], lat[0]))
