#!/bin/bash
# Only the counter passes of tools/gpu_round3.sh (HBM traffic per config, SQ counters of config 5): for a source change that
# leaves the kernels alone (comments) but moves the digest the traffic files are gated on.  usage: tools/gpu_round3_pmc.sh <tag>
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
tools/gpu_pmc_cmd.sh ${TAG}_bench "tcc1 tcc2" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc.log 2>&1; echo "pmc dwt rc=$?"
tools/gpu_pmc_cmd.sh ${TAG}_cfg5 "sq1 sq2 sq3 tcc1 tcc2" -- python bench.py --config cfg5 --steps 2 --warmup 1 > $OUT/pmc_cfg5.log 2>&1; echo "pmc cfg5 rc=$?"
tools/gpu_pmc_cmd.sh ${TAG}_dtcwt "tcc1 tcc2" -- python bench.py --config dtcwt --steps 2 --warmup 1 > $OUT/pmc_dtcwt.log 2>&1; echo "pmc dtcwt rc=$?"
tools/gpu_pmc_cmd.sh ${TAG}_scat "tcc1 tcc2" -- python bench.py --config scat --steps 2 --warmup 1 > $OUT/pmc_scat.log 2>&1; echo "pmc scat rc=$?"
