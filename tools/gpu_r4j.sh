#!/bin/bash
TAG=${1:-r04j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_line.json; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','step_ms_events','timing_consistent','host_issue_ms_per_step')})
print('fwd',d['roofline']['frac'],'inv',d['roofline']['inverse']['frac'])
for k,v in d.get('other_configs',{}).items(): print(k,{a:b for a,b in v.items() if 'kernels' not in a})
for k,v in d.get('other_configs',{}).items():
    if k.startswith('train'): print(k, v['kernels'])
PY
tail -3 $OUT/bench.err
