#!/bin/bash
# A/B builds of the fused-kernel translation unit: tools/build_ab_rows.sh <tag> [extra hipcc flags] -> ab/libwl_<tag>.so (WL_LIB=...)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=pytorch_wavelets_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -fno-slp-vectorize -Wno-inline-asm "$@" -c $C/wl_rows_hip.hip -o ab/rows_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/wl_hip.o ab/rows_$tag.o $C/wl_strip_hip.o $C/wl_dtinv_hip.o -o ab/libwl_$tag.so
echo built ab/libwl_$tag.so
