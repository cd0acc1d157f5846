#!/bin/bash
TAG=${1:-r04f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "changed_after or quadrature or strip" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python tools/gpu_qmf_probe.py 2>&1 | tail -8
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 2> $OUT/bench_cfg5.err | tail -1 > $OUT/bench_cfg5.json
python - <<PY
import json
d=json.load(open('$OUT/bench_cfg5.json'))
r=d['roofline']; print('cfg5 fwd',r['frac'],r['avg_launch_ms'],r['launches']); print('inv',r['inverse']['frac'],r['inverse']['avg_launch_ms'],r['inverse']['launches']); print(r.get('valu'))
PY
