#!/bin/bash
# A/B builds of the main translation unit: tools/build_ab_main.sh <tag> [extra hipcc flags] -> ab/libwl_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=pytorch_wavelets_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC "$@" -c $C/wl_hip.hip -o ab/main_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab/main_$tag.o $C/wl_rows_hip.o $C/wl_strip_hip.o $C/wl_dtinv_hip.o -o ab/libwl_$tag.so
echo built ab/libwl_$tag.so
