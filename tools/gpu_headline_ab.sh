#!/bin/bash
# Same-box A/B of the headline bench with another build of the library (ab/libwl_old.so: the four units of an older commit
# compiled with the flags of __graft_entry__.build()) - is a slower round the box or the code?  (r04j / r04k: the box - 278 500
# Mpixels/s with either library.)
for i in 1 2 3; do for lib in ${WL_AB_LIB:-ab/libwl_old.so} ""; do
WL_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('lib=$lib', d['value'], d['ms_per_step'], 'fwd', r['avg_launch_ms'], 'inv', r['inverse']['avg_launch_ms'], 'tile', r.get('per_level_tile_kernels',{}).get('avg_ms'))"
done; done
