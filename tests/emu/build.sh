#!/bin/bash
# Build the host emulation library (clang from ROCm: needs _Float16 in C++): seven translation units side by side, the
# split of the HIP build (csrc/wl_hip.hip, wl_rows_hip.hip, wl_irows_hip.hip, wl_strip_hip.hip, wl_istrip_hip.hip, wl_dtstrip_hip.hip, wl_dtinv_hip.hip).  A unit is recompiled when one
# of the files IT includes (its compiler-written dependency file) is newer than its object.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-O2 -g -std=c++17 -fno-strict-aliasing -fPIC -fopenmp -Wall -Wno-unused-function $WL_EMU_EXTRA"
pids=()
stamp=$(mktemp); touch $stamp    # objects get the time the build STARTED: a source edited meanwhile stays newer
built=()
for u in api rows irows strip istrip dtstrip dtinv; do
  stale=0
  if [ ! -f wl_emu_$u.o ] || [ ! -f wl_emu_$u.o.d ] || [ build.sh -nt wl_emu_$u.o ]; then stale=1; else
    for d in $(sed -e 's/\\$//' -e 's/^[^:]*://' wl_emu_$u.o.d); do
      case $d in /opt/*|/usr/*) continue;; esac
      if [ ! -f "$d" ] || [ "$d" -nt wl_emu_$u.o ]; then stale=1; break; fi
    done
  fi
  if [ $stale = 1 ]; then
    $CXX $FLAGS -MD -MF wl_emu_$u.o.d -c wl_emu_$u.cpp -o wl_emu_$u.o & pids+=($!); built+=(wl_emu_$u.o)
  fi
done
for p in "${pids[@]}"; do wait $p; done
for o in "${built[@]}"; do touch -r $stamp $o; done
$CXX -shared -fopenmp wl_emu_api.o wl_emu_rows.o wl_emu_irows.o wl_emu_strip.o wl_emu_istrip.o wl_emu_dtstrip.o wl_emu_dtinv.o -o libwl_emu.so
touch -r $stamp libwl_emu.so; rm -f $stamp
echo built tests/emu/libwl_emu.so
