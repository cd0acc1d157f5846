#!/bin/bash
# Round 3, GPU call A: parity tests, the bench line (+ membench), non-temporal A/B of the streaming kernels, the other configs.
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
for v in base nt1 nt2 nt3 base nt1 nt2 nt3; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>>$OUT/ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print(json.dumps({'lib': '$v', 'ms_per_step': d['ms_per_step'], 'cold_ms': d['cold']['ms_per_step'], 'fwd_ms': r['avg_launch_ms'], 'fwd_frac': r['frac'], 'inv_ms': r['inverse']['avg_launch_ms'], 'inv_frac': r['inverse']['frac'], 'closure': r['closure'], 'copy': r.get('device_copy_gbs')}))" >> $OUT/ab.jsonl
done
cat $OUT/ab.jsonl
for c in dtcwt scat cfg5; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2>> $OUT/bench.err; echo "bench $c rc=$?"
done
rocm-smi --showclocks --showpower > $OUT/box.txt 2>&1; lscpu | head -20 >> $OUT/box.txt
head -c 6000 $OUT/bench_line.json
