#!/bin/bash
# Build the host emulation library (clang from ROCm: needs _Float16 in C++).
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -O2 -g -std=c++17 -fno-strict-aliasing -fPIC -shared -fopenmp -Wall -Wno-unused-function wl_emu.cpp -o libwl_emu.so
echo built tests/emu/libwl_emu.so
