#!/bin/bash
# Round 6 GPU call: parity tests, [PMC=1: the counter passes first - HBM traffic per config, SQ counters of config 5 -] the bench
# line of every config, rocprofv3 kernel traces.  usage (on the GPU box, from the repo root): [PMC=1] tools/gpu_round6.sh <tag>
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
if [ "${PMC:-0}" = "1" ]; then
  tools/gpu_pmc_cmd.sh ${TAG}_bench "tcc1 tcc2" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc.log 2>&1; echo "pmc dwt rc=$?"
  tools/gpu_pmc_cmd.sh ${TAG}_cfg5 "sq1 sq2 sq3 tcc1 tcc2" -- python bench.py --config cfg5 --steps 2 --warmup 1 > $OUT/pmc_cfg5.log 2>&1; echo "pmc cfg5 rc=$?"
  tools/gpu_pmc_cmd.sh ${TAG}_dtcwt "tcc1 tcc2" -- python bench.py --config dtcwt --steps 2 --warmup 1 > $OUT/pmc_dtcwt.log 2>&1; echo "pmc dtcwt rc=$?"
  tools/gpu_pmc_cmd.sh ${TAG}_scat "tcc1 tcc2" -- python bench.py --config scat --steps 2 --warmup 1 > $OUT/pmc_scat.log 2>&1; echo "pmc scat rc=$?"
  # the traffic files of THIS build, so that the bench lines below quote roofline.traffic (digest-gated)
  python tools/make_round_profiles.py $TAG r06 traffic > $OUT/traffic.log 2>&1; echo "traffic files rc=$?"; cp profiles/r06_*traffic*.json profiles/r06_cfg5_pmc_summary.json $OUT/ 2>/dev/null
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_20.json 2> $OUT/bench.err; echo "bench(20/5) rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_line.json 2>> $OUT/bench.err; echo "bench rc=$?"
for c in dtcwt scat cfg5; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>> $OUT/bench.err | tail -1 > $OUT/bench_$c.json; echo "bench $c rc=$?"
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $REPO/$OUT/prof.log 2>&1); echo "rocprof dwt rc=$?"
for c in dtcwt scat cfg5; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$c -o bench -- python $REPO/bench.py --config $c --steps 10 --warmup 3 > $REPO/$OUT/prof_$c.log 2>&1); echo "rocprof $c rc=$?"
done
rocm-smi --showclocks --showpower > $OUT/box.txt 2>&1; lscpu | head -20 >> $OUT/box.txt
head -c 1500 $OUT/bench_line_20.json
