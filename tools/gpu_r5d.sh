#!/bin/bash
# Round 5, GPU call D: parity tests, same-box A/B (old package | this one), the rocprofv3 kernel trace of the A/B run (the
# one-thread examination kernel, the armed fallbacks), bench lines of the metric and of config 5.
TAG=${1:-r05d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for i in 1 2; do
  WL_PKG_ROOT=ab/old_pkg timeout 600 python tools/gpu_r5_ab.py old 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-200
  timeout 600 python tools/gpu_r5_ab.py new 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-200
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_lat -o ab -- python $REPO/tools/gpu_r5_ab.py prof > $REPO/$OUT/prof_lat.log 2>&1); echo "rocprof rc=$?"
f=$(find $OUT/prof_lat -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_lat.csv && grep -E "TapPrep|Strip" $f | cut -c1-160
rm -rf $OUT/prof_lat
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 2> $OUT/bench.err | tail -1 > $OUT/bench_cfg5.json; echo "bench cfg5 rc=$?"; cut -c1-1500 $OUT/bench_cfg5.json
