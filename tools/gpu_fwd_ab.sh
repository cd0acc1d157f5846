#!/bin/bash
# same-box A/B of the fused / lean DTCWT forward kernels and ScatLayer: variants built by tools/build_ab_strip.sh
OUT=gpurun_out/fwdab; mkdir -p $OUT
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
    x5 = torch.randn(64, 3, 512, 512, device=dev); x2 = torch.randn(256, 3, 256, 256, device=dev)
    for tag, m, x in (('dtcwt_J3_512', pw.DTCWTForward(J=3).to(dev), x5), ('dtcwt_J1_512', pw.DTCWTForward(J=1).to(dev), x5),
                      ('dtcwt_J2_256', pw.DTCWTForward(J=2).to(dev), x2), ('scat_512', pw.ScatLayer().to(dev), x5), ('scat_256', pw.ScatLayer().to(dev), x2)):
        m(x)
        out[tag] = round(timeit(lambda: m(x)), 4)
print(json.dumps(out))
PY
done
