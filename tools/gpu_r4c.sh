#!/bin/bash
# Round 4: A/B of the fused inverse's prefetch depth on one box (bench --config dtcwt, events).
TAG=${1:-r04c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "" ab/libwl_pf22.so ab/libwl_pf84.so ""; do
  WL_LIB=$lib timeout 600 python bench.py --config dtcwt --steps 20 --warmup 5 2>> $OUT/bench.err | tail -1 > $OUT/b.json
  python - <<PY
import json
d=json.load(open('$OUT/b.json'))
r=d['roofline']; print('$lib', 'step', d['ms_per_step'], 'fwd',r['frac'],r['avg_launch_ms'], 'inv',r['inverse']['frac'],r['inverse']['avg_launch_ms'],r['inverse']['launches'])
PY
done
