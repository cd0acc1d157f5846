#!/bin/bash
# same-box A/B of the level >= 2 DTCWT tile kernels' tile shapes: variants built by tools/build_ab_main.sh
OUT=gpurun_out/l2ab; mkdir -p $OUT
for v in base "$@"; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L timeout 200 python - <<PY 2>>$OUT/err.log | tee -a $OUT/ab.jsonl
import json, os, sys, torch
sys.path.insert(0, '.')
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
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
xfm = pw.DTCWTForward(J=3).to(dev); ifm = pw.DTCWTInverse().to(dev)
with torch.no_grad():
    x = torch.randn(64, 3, 512, 512, device=dev)
    yl, yh = xfm(x)
    ll1 = torch.randn(64, 3, 512, 512, device=dev)
    ll2 = torch.randn(64, 3, 256, 256, device=dev)
    out['fwd2_512'] = round(timeit(lambda: ops.dtcwt_fwd2(ll1, xfm.h0a, xfm.h0b, xfm.h1a, xfm.h1b)), 4)
    out['fwd2_256'] = round(timeit(lambda: ops.dtcwt_fwd2(ll2, xfm.h0a, xfm.h0b, xfm.h1a, xfm.h1b)), 4)
    out['inv2_to512'] = round(timeit(lambda: ops.dtcwt_inv2(ll2, yh[1], ifm.g0a, ifm.g0b, ifm.g1a, ifm.g1b)), 4)
    l3 = torch.randn(64, 3, 128, 128, device=dev)
    out['inv2_to256'] = round(timeit(lambda: ops.dtcwt_inv2(l3, yh[2], ifm.g0a, ifm.g0b, ifm.g1a, ifm.g1b)), 4)
    out['fwd_J3'] = round(timeit(lambda: xfm(x)), 4)
    out['inv_J3'] = round(timeit(lambda: ifm((yl, yh))), 4)
    r = ifm((yl, yh)); out['rt_err'] = float((r - x).abs().max() / x.abs().max())
print(json.dumps(out))
PY
done
