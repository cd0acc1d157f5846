#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
# usage: tools/gpu_check.sh <tag> [pytest-args...]
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > "$OUT/box.txt" 2>&1
echo "== smoke" | tee "$OUT/smoke.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x "$@" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -15 "$OUT/pytest_gpu.log"
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== rocprofv3 kernel stats"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?"
find "$OUT/prof" -name "*kernel_stats*" | head -3
for f in $(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1); do head -12 "$f"; done
