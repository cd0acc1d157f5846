#!/bin/bash
# same-box A/B: product library vs A/B builds of the strip translation unit (tools/gpu_r5_cw.py); LIBS="ab/libwl_x.so ..." 
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "" ${LIBS:-ab/libwl_cw2.so ab/libwl_cw1.so}; do
    WL_LIB=$lib timeout 300 python tools/gpu_r5_cw.py 2>&1 | tail -1 >> gpurun_out/r5_cw.jsonl
  done
done
python - <<'P'
import json
for l in open('gpurun_out/r5_cw.jsonl'):
    d = json.loads(l)
    print(d['lib'] or 'product', {k: v for k, v in d.items() if k.startswith('h16per') and 'kern' not in k})
P
