#!/bin/bash
# same-box A/B of the lean ScatLayer kernel: variants built by tools/build_ab_strip.sh
OUT=gpurun_out/scatab; mkdir -p $OUT
for v in base "$@"; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L timeout 200 python - <<PY 2>>$OUT/err.log | tee -a $OUT/ab.jsonl
import json, os, sys, torch
sys.path.insert(0, '.')
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[1]
out = {'lib': '$v'}
with torch.no_grad():
    for shape in ((256, 3, 256, 256), (64, 3, 512, 512), (16, 3, 1024, 1024)):
        m = pw.ScatLayer().to(dev)
        x = torch.randn(*shape, device=dev)
        m(x)
        out['x'.join(map(str, shape))] = round(timeit(lambda: m(x)), 4)
    out['kernel'] = pw.last_kernel()
print(json.dumps(out))
PY
done
