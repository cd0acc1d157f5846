#!/bin/bash
# Round 3, GPU call B: the full GPU test suite, then the non-temporal A/B of the two streaming kernels (same box).
OUT=gpurun_out/r03b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for v in base nt1 nt2 nt3 base nt1 nt2 nt3; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>>$OUT/ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print(json.dumps({'lib': '$v', 'ms_per_step': d['ms_per_step'], 'cold_ms': d['cold']['ms_per_step'], 'fwd_ms': r['avg_launch_ms'], 'fwd_frac': r['frac'], 'inv_ms': r['inverse']['avg_launch_ms'], 'inv_frac': r['inverse']['frac'], 'closure': r['closure'], 'copy': r.get('device_copy_gbs')}))" >> $OUT/ab.jsonl
done
cat $OUT/ab.jsonl
