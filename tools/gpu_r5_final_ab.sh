#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r5_final_ab.jsonl
for rep in 1 2; do
  WL_PKG_ROOT=ab/old_pkg timeout 600 python tools/gpu_r5_final_ab.py 2>/dev/null | tail -1 >> gpurun_out/r5_final_ab.jsonl
  timeout 600 python tools/gpu_r5_final_ab.py 2>/dev/null | tail -1 >> gpurun_out/r5_final_ab.jsonl
done
python - <<'P'
import json
rows = [json.loads(l) for l in open('gpurun_out/r5_final_ab.jsonl')]
old = [r for r in rows if r['pkg'] != 'round5']; new = [r for r in rows if r['pkg'] == 'round5']
for k in new[0]:
    if k == 'pkg': continue
    o = min(r[k] for r in old); n = min(r[k] for r in new)
    print('%-34s round 4 %.4f ms   round 5 %.4f ms   %+.1f %%' % (k, o, n, 100 * (n - o) / o))
P
