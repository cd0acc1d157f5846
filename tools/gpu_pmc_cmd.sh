#!/bin/bash
# rocprofv3 counter passes (counters + kernel trace only) around an arbitrary command.
# usage: tools/gpu_pmc_cmd.sh <tag> <passes: e.g. "sq1 sq2 sq3 tcc1 tcc2"> -- <command ...>
TAG=$1; PASSES=$2; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
declare -A C
C[sq1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
C[sq2]="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C[sq3]="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
C[tcc1]="FETCH_SIZE"
C[tcc2]="WRITE_SIZE"
for p in $PASSES; do
  (cd "$REPO" && timeout 600 rocprofv3 --pmc ${C[$p]} --kernel-trace --output-format csv -d "$OUT/$p" -o p -- "$@" > "$OUT/$p.log" 2>&1)
  echo "pass $p rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(out + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
summary = {}
for k in vals:
    if 'wl_kernel' not in k: continue
    short = k.replace('void ', '').split('(')[0].replace(' >', '>')
    summary[short] = {}
    for c, d in sorted(vals[k].items()):
        v = sorted(d.values())
        summary[short][c] = {'dispatches': len(v), 'mean': sum(v) / len(v), 'max': v[-1]}
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
