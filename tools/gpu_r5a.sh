#!/bin/bash
# Round 5, GPU call A: smoke + parity tests on the guarded kernels, then the same-box A/B of the previous round's library
# (ab/libwl_old.so) against this one on the launches the tap-relation guards touch (tools/gpu_r5_ab.py).
TAG=${1:-r05a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for i in 1 2; do
  WL_PKG_ROOT=ab/old_pkg timeout 600 python tools/gpu_r5_ab.py old 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl
  WL_NO_LATTICE=1 timeout 600 python tools/gpu_r5_ab.py new_qmf 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl
  timeout 600 python tools/gpu_r5_ab.py new 2>> $OUT/ab.err | tail -1 | tee -a $OUT/ab.jsonl
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench.err; echo "bench(20/5) rc=$?"
head -c 1200 $OUT/bench_line_20.json
