#!/bin/bash
# Round 5, GPU call K: parity tests, then the headline bench with the product library and with the A/B build that runs the lattice
# variants at 8 taps as well (ab/libwl_lat8.so, ops.ROWS_LATTICE_MIN = 8), interleaved, same box
TAG=${1:-r05k}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for i in 1 2 3; do
  timeout 600 python tools/gpu_r5_ab.py new 2>> $OUT/ab.err | tail -1 >> $OUT/ab.jsonl
  WL_LIB=ab/libwl_lat8.so WL_ROWS_LAT8=1 timeout 600 python tools/gpu_r5_ab.py lat8 2>> $OUT/ab.err | tail -1 >> $OUT/ab.jsonl
done
python - $OUT <<'PY'
import json, sys
rows=[json.loads(l) for l in open(sys.argv[1] + '/ab.jsonl')]
keys=[k for k in rows[0] if not k.endswith('_k') and k!='lib']
print('%-14s'%'case', *['%-9s'%r['lib'][-9:] for r in rows])
for k in keys[:8]:
    print('%-14s'%k, *['%-9s'%r.get(k) for r in rows])
print(rows[1].get('fwd_db4_k'), rows[1].get('inv_db4_k'))
PY
