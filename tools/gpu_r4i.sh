#!/bin/bash
for lib in ab/libwl_s3k.so ab/libwl_s4k.so ab/libwl_s6k.so ab/libwl_s8k.so; do
  echo "== $lib"; WL_LIB=$lib timeout 300 python tools/gpu_dwt1d_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
