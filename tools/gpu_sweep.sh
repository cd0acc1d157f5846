#!/bin/bash
# Sweep the streaming-kernel knobs on the GPU box: prints one compact line per setting.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"; mkdir -p gpurun_out/sweep
for S in 1 2 3 4; do for RS in 8 4; do for KB in 78 52 158; do
  WL_STREAM_STRIPS=$S WL_STREAM_RS=$RS WL_STREAM_LDS_KB=$KB timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('S=$S RS=$RS KB=$KB fwd_ms=%.4f frac=%.3f inv_ms=%.4f value=%.0f err=%.1e'%(r['avg_launch_ms'], r['frac'], r['inverse']['avg_ms'], d['value'], d['roundtrip_rel_err']))"
done; done; done | tee gpurun_out/sweep/sweep.txt
