#!/bin/bash
# A/B builds of the strip translation unit for same-box measurements: tools/build_ab_strip.sh <tag> [extra hipcc flags]
# -> ab/libwl_<tag>.so (load with WL_LIB=ab/libwl_<tag>.so).  The other two objects are reused from the product build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=pytorch_wavelets_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -fno-slp-vectorize -Wno-inline-asm "$@" -c $C/wl_strip_hip.hip -o ab/strip_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/wl_hip.o $C/wl_rows_hip.o ab/strip_$tag.o $C/wl_dtinv_hip.o -o ab/libwl_$tag.so
echo built ab/libwl_$tag.so
