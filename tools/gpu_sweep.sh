#!/bin/bash
# Sweep the streaming-kernel knobs on the GPU box: prints one compact line per setting.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"; mkdir -p gpurun_out/sweep
run() {
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*'.ljust(70), 'fwd_ms=%.4f frac=%.3f inv_ms=%.4f value=%.0f err=%.1e'%(r['avg_launch_ms'], r['frac'], r['inverse']['avg_ms'], d['value'], d['roundtrip_rel_err']))"
}
for cfg in "$@"; do run $cfg; done | tee -a gpurun_out/sweep/sweep.txt
