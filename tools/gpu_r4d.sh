#!/bin/bash
for lib in ab/libwl_v2.so ab/libwl_v2pf3.so ab/libwl_v2ko32.so ab/libwl_v2ko63.so ab/libwl_v2ko1.so ab/libwl_v2ko16.so ab/libwl_v2.so; do
  WL_LIB=$lib timeout 120 python tools/gpu_dti21_time.py 2>&1 | tail -1
done
WL_LIB=ab/libwl_v2.so timeout 120 python tools/gpu_dti21_time.py 16 3 1024 1024 2>&1 | tail -1
WL_LIB=ab/libwl_v2.so timeout 120 python tools/gpu_dti21_time.py 256 3 256 256 2>&1 | tail -1
