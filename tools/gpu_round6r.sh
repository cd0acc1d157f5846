#!/bin/bash
# Round 6, last call: the round's complete set (tools/gpu_round6.sh with the counter passes) + the near_sym_b / qshift_d / near_sym_b_bp
# evidence on the same box: A/B against the tile kernels, rocprofv3 kernel durations of that probe, the DTCWT fuzzers.
# usage (GPU box, repo root): tools/gpu_round6r.sh <tag>
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
PMC=1 tools/gpu_round6.sh $TAG
timeout 500 python tools/gpu_r6_nsb.py > $OUT/nsb_ab.jsonl 2> $OUT/nsb_ab.err; echo "nsb A/B rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_nsb -o nsb -- python $REPO/tools/gpu_r6_nsb.py > $REPO/$OUT/prof_nsb.log 2>&1); echo "rocprof nsb rc=$?"
timeout 600 python tools/gpu_round6c_fuzz.py 1 60 > $OUT/fuzz_nsb.txt 2>&1; echo "fuzz nsb rc=$?"; tail -1 $OUT/fuzz_nsb.txt
timeout 600 python tools/gpu_dtcwt_fuzz.py 3 40 > $OUT/fuzz_dtcwt.txt 2>&1; echo "fuzz dtcwt rc=$?"; tail -1 $OUT/fuzz_dtcwt.txt
timeout 600 python tools/gpu_narrow_dtcwt_fuzz.py 2 > $OUT/fuzz_narrow_dtcwt.txt 2>&1; echo "fuzz narrow rc=$?"; tail -1 $OUT/fuzz_narrow_dtcwt.txt
