#!/bin/bash
TAG=${1:-r04g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python tools/gpu_dwt1d_time.py 2>&1 | tail -6
