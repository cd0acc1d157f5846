"""Round 5: does the DTCWT forward's time depend on WHERE the caching allocator places its tensors?  Random junk allocations (kept
alive) before each trial move x and the outputs around; addresses (MB) and times are printed."""
import json, os, sys, torch, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
def t(fn, n=40):
    with torch.no_grad():
        fn(); fn()
        return round(min(bench.time_seq_fn(fn, n, sync) for _ in range(4)), 4)
d = pw.DTCWTForward(J=3).to(dev)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
MB = 1 << 20
junk = []
for trial in range(14):
    if trial:
        for _ in range(random.randint(1, 4)):
            junk.append(torch.empty(random.randint(1, 700) * MB + random.choice([0, 512, 4096, 1 << 16]), dtype=torch.uint8, device=dev))
        if random.random() < 0.5 and junk:
            del junk[random.randrange(len(junk))]
    x = torch.randn(64, 3, 512, 512, device=dev)
    with torch.no_grad():
        yl, yh = d(x)
    ptrs = [x.data_ptr()] + [yl.data_ptr()] + [h.data_ptr() for h in yh]
    row = {'trial': trial, 'ms': t(lambda: d(x)), 'ptr_MB': [round(p / MB, 2) for p in ptrs], 'rel_MB': [round((p - ptrs[0]) / MB, 2) for p in ptrs[1:]]}
    print(json.dumps(row), flush=True)
    del x, yl, yh
