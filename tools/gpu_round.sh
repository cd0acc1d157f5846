#!/bin/bash
# One GPU call of a round: parity tests, the bench line, rocprofv3 kernel trace + PMC traffic of the bench command.
# usage (on the GPU box, from the repo root): tools/gpu_round.sh <tag>
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $REPO/$OUT/prof.log 2>&1); echo "rocprof rc=$?"
tools/gpu_pmc_cmd.sh ${TAG}_bench "tcc1 tcc2" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
rocm-smi --showclocks --showpower > $OUT/box.txt 2>&1; lscpu | head -20 >> $OUT/box.txt
cat $OUT/bench_line.json | head -c 3000
