#!/bin/bash
# A/B builds of the engine for same-box measurements: tools/build_ab.sh <tag> [extra hipcc flags for the rows TU]
# -> ab/libwl_<tag>.so (load with WL_LIB=ab/libwl_<tag>.so).  The main TU object is reused from the product build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
C=pytorch_wavelets_amd/csrc
mkdir -p ab
[ -f $C/wl_hip.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -c $C/wl_hip.hip -o $C/wl_hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC -fno-slp-vectorize -Wno-inline-asm "$@" -c $C/wl_rows_hip.hip -o ab/rows_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/wl_hip.o ab/rows_$tag.o -o ab/libwl_$tag.so
echo built ab/libwl_$tag.so
