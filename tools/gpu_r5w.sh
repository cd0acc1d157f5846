#!/bin/bash
# same-box A/B: pack-and-cut of few narrow planes in the fused kernels (tools/gpu_r5w.py), product vs -DWL_ROWS_PACK_CUT=0
for rep in 1 2; do for lib in "" ab/libwl_nopack.so; do
  echo "== ${lib:-product}"; WL_LIB=$lib python tools/gpu_r5w.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], d['J'], 'fwd', d['fwd_ms'], d['fwd_frac'], d['gf'], 'inv', d['inv_ms'], d['inv_frac'], d['gi'], 'inv(lattice)', d['inv_ms_lattice'])"
done; done
