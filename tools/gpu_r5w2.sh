#!/bin/bash
# same-box A/B: one loader for the finest level's bands on narrow multi-level synthesis (product) vs -DWL_IROWS_NARROW_ML=0
for rep in 1 2; do for lib in "" ${LIB2:-ab/libwl_nonml.so}; do
  echo "== ${lib:-product}"; WL_LIB=$lib python tools/gpu_r5w.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['J'] > 1: print(d['shape'], d['J'], 'inv', d['inv_ms'], d['inv_frac'], d['gi'], 'inv(lattice)', d['inv_ms_lattice'])"
done; done
