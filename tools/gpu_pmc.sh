#!/bin/bash
# rocprofv3 PMC passes for the bench (counters only + kernel trace, as the pool requires).
# usage: tools/gpu_pmc.sh <tag> [ENV=VAL ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  env "${ENVV[@]}" timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
ENVV=("$@")
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
pass sq2 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
python - "$OUT" "$REPO" <<'PY'
import csv, glob, json, os, sys, collections
out, repo = sys.argv[1], sys.argv[2]
# per-kernel: list of per-dispatch counter values (dispatch order)
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(out + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
summary, traffic = {}, {}
for k in vals:
    if 'wl_kernel' not in k: continue
    short = k.replace('void ', '').split('(')[0].replace(' >', '>')
    print('==', short)
    summary[short] = {}
    for c, d in sorted(vals[k].items()):
        v = sorted(d.values())
        summary[short][c] = {'dispatches': len(v), 'mean': sum(v) / len(v), 'max': v[-1]}
        print('   %-28s n=%3d  mean %16.0f  max %16.0f' % (c, len(v), sum(v) / len(v), v[-1]))
    if 'FETCH_SIZE' in vals[k] and 'WRITE_SIZE' in vals[k]:
        # the largest dispatch is the level-1 launch.  Units: KB.  gfx950 correction (MI355X_MICROARCH.md, HBM):
        # FETCH_SIZE reports half the bytes of a wide coalesced read stream -> doubled; WRITE_SIZE as is.
        fs, ws = max(vals[k]['FETCH_SIZE'].values()), max(vals[k]['WRITE_SIZE'].values())
        traffic[short] = {'fetch_bytes_raw': fs * 1024, 'write_bytes_raw': ws * 1024,
                          'hbm_bytes_corrected': 2 * fs * 1024 + ws * 1024}
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1)
json.dump(traffic, open(out + '/hbm_traffic.json', 'w'), indent=1)
print(json.dumps(traffic))
PY
