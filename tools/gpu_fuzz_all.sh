#!/bin/bash
# Every GPU fuzzer of rounds 2-6 on the current build, one file: usage (GPU box, repo root): tools/gpu_fuzz_all.sh <tag>
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
F=$OUT/fuzz_all.txt; : > $F
run() { echo "== $1 $2 $3" >> $F; timeout 900 python tools/$1.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -${4:-3} | cut -c1-600 >> $F; echo "$1 rc=$?"; }
run gpu_fuzz 200 "" 1
run gpu_dtcwt_fuzz 11 60 2
run gpu_round4_fuzz 300 "" 1
run gpu_round5_fuzz 300 "" 1
run gpu_round5b_fuzz 240 "" 2
run gpu_round6_fuzz 300 "" 2
run gpu_round6b_fuzz 200 "" 2
run gpu_round6c_fuzz 6 80 2
run gpu_narrow_dtcwt_fuzz 60 "" 1
grep -c "FAILURES: 0\|\"bad\": 0\|0 mismatches\|\"nbad\": 0" $F
