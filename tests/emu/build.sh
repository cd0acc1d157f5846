#!/bin/bash
# Build the host emulation library (clang from ROCm: needs _Float16 in C++): four translation units side by side, the
# split of the HIP build (csrc/wl_hip.hip, wl_rows_hip.hip, wl_strip_hip.hip, wl_dtinv_hip.hip).
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-O2 -g -std=c++17 -fno-strict-aliasing -fPIC -fopenmp -Wall -Wno-unused-function"
pids=()
for u in api rows strip dtinv; do
  $CXX $FLAGS -c wl_emu_$u.cpp -o wl_emu_$u.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -fopenmp wl_emu_api.o wl_emu_rows.o wl_emu_strip.o wl_emu_dtinv.o -o libwl_emu.so
echo built tests/emu/libwl_emu.so
