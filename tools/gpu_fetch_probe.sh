#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the DWT launches under an env setting.  usage: tools/gpu_fetch_probe.sh TAG [ENV=VAL...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/fetch_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > "$OUT/$c.log" 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, sys
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'wl_kernel' in r['Kernel_Name']:
            v[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(v):
    print(sys.argv[2], k, ' '.join('%s max %.1f MB' % (c, max(x) * 1024 / 1e6) for c, x in sorted(v[k].items())))
PY
