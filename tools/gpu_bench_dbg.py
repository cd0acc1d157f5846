"""GPU probe: replay bench.py's cold / ramp / steady flow for --config dtcwt with per-step wall-clock stamps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
xfm, ifm = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
x = torch.randn(64, 3, 512, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
state = {}
def step():
    with torch.no_grad():
        state['c'] = xfm(x)
        state['r'] = ifm(state['c'])
def region(tag, W, K):
    for _ in range(W): step()
    torch.cuda.synchronize()
    stamps = [time.perf_counter()]
    for _ in range(K):
        step(); stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(tag, 'total %.3f ms/step; host per step:' % ((t1 - stamps[0]) / K * 1e3), ' '.join('%.2f' % ((b - a) * 1e3) for a, b in zip(stamps, stamps[1:])), flush=True)
region('cold', 3, 10)
t0 = time.perf_counter()
for _ in range(100): step()
print('ramp issue %.1f ms' % ((time.perf_counter() - t0) * 1e3))
region('steady', 3, 10)
region('again', 3, 10)
print(torch.cuda.memory_stats()['num_alloc_retries'], torch.cuda.memory_stats()['reserved_bytes.all.current'] / 1e9)
