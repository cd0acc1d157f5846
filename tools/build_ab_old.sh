#!/bin/bash
# Same-box A/B against the library of the last commit: the four translation units of ab/old_src (git archive HEAD ...) compiled
# with the flags of __graft_entry__.build() -> ab/libwl_old.so (load with WL_LIB=ab/libwl_old.so).
set -e
cd "$(dirname "$0")/.."
C=ab/old_src/pytorch_wavelets_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -fPIC"
/opt/rocm/bin/hipcc $F -c $C/wl_hip.hip -o ab/old_main.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -Wno-inline-asm -c $C/wl_rows_hip.hip -o ab/old_rows.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -Wno-inline-asm -c $C/wl_strip_hip.hip -o ab/old_strip.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -Wno-inline-asm -c $C/wl_dtinv_hip.hip -o ab/old_dtinv.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab/old_main.o ab/old_rows.o ab/old_strip.o ab/old_dtinv.o -o ab/libwl_old.so
echo built ab/libwl_old.so
