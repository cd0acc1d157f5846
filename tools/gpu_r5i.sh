#!/bin/bash
# Round 5, GPU call I: parity tests, then the same-box A/B (old package | this one)
TAG=${1:-r05i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for i in 1 2; do
  WL_PKG_ROOT=ab/old_pkg timeout 600 python tools/gpu_r5_ab.py old 2>> $OUT/ab.err | tail -1 >> $OUT/ab.jsonl
  timeout 600 python tools/gpu_r5_ab.py new 2>> $OUT/ab.err | tail -1 >> $OUT/ab.jsonl
done
python - $OUT <<'PY'
import json, sys
rows=[json.loads(l) for l in open(sys.argv[1] + '/ab.jsonl')]
keys=[k for k in rows[1] if not k.endswith('_k') and k!='lib']
print('%-14s'%'case', *['%-9s'%r['lib'][-9:] for r in rows])
for k in keys:
    print('%-14s'%k, *['%-9s'%r.get(k) for r in rows])
print(rows[-1].get('fwd_db4_k'), rows[1].get('fwd_db8_k'))
PY
