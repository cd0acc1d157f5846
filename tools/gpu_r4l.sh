#!/bin/bash
TAG=${1:-r04l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dtcwt_gpu.py -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for lib in "" ab/libwl_nopairs.so ""; do WL_LIB=$lib timeout 300 python tools/gpu_scat_time.py 2>&1 | grep -v amdgpu.ids; done
