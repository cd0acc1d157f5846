#!/bin/bash
# (round 4.  The python-side knobs STRIP_MINW_F16 / ISTRIP_MINW_F16 became ops.STRIP_MINW in round 5 - the launchers decide the rest - so the
# attribute assignment below no longer changes anything; the library-side macros of the A/B builds still do.)
# float16 levels narrower than 2 KiB / 1 KiB rows on the strip kernels?  A/B builds ab/libwl_a<W>[i<W>].so (tools/build_ab_strip.sh
# <tag> -DWL_STRIP_MINW=<W> [-DWL_ISTRIP_MINW=<W>]) + the python-side mirror of the rule, config 5.
for v in "0 0 " "512 0 ab/libwl_a512.so" "256 0 ab/libwl_a256.so" "256 256 ab/libwl_a256i256.so" "512 256 ab/libwl_a512i256.so" "0 0 "; do
set -- $v
WL_LIB=$3 python -c "
import sys, json, io, contextlib
import pytorch_wavelets_amd.ops as o
o.STRIP_MINW_F16, o.ISTRIP_MINW_F16 = $1, $2
import bench
sys.argv = ['bench.py', '--config', 'cfg5', '--steps', '10', '--warmup', '3']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1]); r = d['roofline']
print('$1 $2', d['ms_per_step'], 'fwd', r.get('frac'), r.get('avg_launch_ms'), 'inv', r.get('inverse', {}).get('frac'), r.get('launches'))" 2>/dev/null
done
