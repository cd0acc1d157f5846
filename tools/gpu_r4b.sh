#!/bin/bash
# Round 4: parity tests + the DTCWT bench line (fused inverse) + its kernel trace.
TAG=${1:-r04b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --config dtcwt --steps 20 --warmup 5 2> $OUT/bench_dtcwt.err | tail -1 > $OUT/bench_dtcwt.json; echo "bench dtcwt rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench_dtcwt.json'))
print({k:d[k] for k in ('value','ms_per_step','step_ms_events','timing_consistent','host_issue_ms_per_step')})
r=d['roofline']; print('fwd',r['frac'],r['avg_launch_ms'],r['launches']); print('inv',r['inverse']['frac'],r['inverse']['avg_launch_ms'],r['inverse']['launches'])
PY
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_dtcwt -o bench -- python $REPO/bench.py --config dtcwt --steps 20 --warmup 5 > $REPO/$OUT/prof_dtcwt.log 2>&1); echo "rocprof dtcwt rc=$?"
f=$(find $OUT/prof_dtcwt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220
