#!/bin/bash
TAG=${1:-r04m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python tools/gpu_scat_time.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/gpu_dti21_time.py 256 3 256 256 2>&1 | tail -1
