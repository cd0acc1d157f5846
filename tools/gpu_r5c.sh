#!/bin/bash
# Round 5, GPU call C: same-box A/B of (old package | QMF variant | lattice variant | lattice at six waves per SIMD), then the
# rocprofv3 kernel traces of the lattice and the QMF runs (per-kernel durations: WlTapPrep, the armed fallbacks).
TAG=${1:-r05c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
for i in 1 2; do
  WL_PKG_ROOT=ab/old_pkg timeout 600 python tools/gpu_r5_ab.py old 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-300
  WL_NO_LATTICE=1 timeout 600 python tools/gpu_r5_ab.py new_qmf 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-300
  timeout 600 python tools/gpu_r5_ab.py new 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-300
  WL_LIB=ab/libwl_mw6.so timeout 600 python tools/gpu_r5_ab.py mw6 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl | cut -c1-300
done
for v in lat qmf; do
  (cd /tmp && export TMPDIR=/tmp && WL_NO_LATTICE=$([ $v = qmf ] && echo 1) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$v -o ab -- python $REPO/tools/gpu_r5_ab.py prof_$v > $REPO/$OUT/prof_$v.log 2>&1); echo "rocprof $v rc=$?"
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$v.csv && head -40 $f | cut -c1-200
done
