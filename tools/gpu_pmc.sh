#!/bin/bash
# rocprofv3 PMC passes for the bench (counters only + kernel trace, as the pool requires).
# usage: tools/gpu_pmc.sh <tag> [ENV=VAL ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  env "${ENVV[@]}" timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
ENVV=("$@")
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
pass sq2 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/*/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen.add((k, r['Dispatch_Id']))
    for k, _ in seen: cnt[(f, k)] += 1
for k in agg:
    if 'wl_kernel' not in k: continue
    print('==', k)
    for c, v in sorted(agg[k].items()):
        n = max(cnt[(f, k)] for f in glob.glob(out + '/*/*counter_collection.csv'))
        print('   %-28s %16.0f  per-dispatch %14.0f' % (c, v, v / max(n, 1)))
PY
