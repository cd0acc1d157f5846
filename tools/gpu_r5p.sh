#!/bin/bash
# round 5: planes-per-workgroup on narrow strip levels - parity tests, same-box A/B against the one-plane build, config 5
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dwt_gpu.py -x -q -m gpu -k "several_planes or strip or lattice or periodization or float16" 2>&1 | tail -5 > gpurun_out/r5p_pytest.log
rm -f gpurun_out/r5_cw.jsonl
LIBS="ab/libwl_pp1.so" tools/gpu_r5_cw.sh > gpurun_out/r5p_ab.log 2>&1
for lib in "" ab/libwl_pp1.so "" ab/libwl_pp1.so; do
  WL_LIB=$lib timeout 600 python bench.py --config cfg5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib' or 'product', d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['inverse']['frac'], r['inverse']['avg_launch_ms'])" >> gpurun_out/r5p_cfg5.log
done
cat gpurun_out/r5p_pytest.log gpurun_out/r5p_ab.log gpurun_out/r5p_cfg5.log
