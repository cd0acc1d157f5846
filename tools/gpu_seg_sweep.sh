python -m pytest tests/test_dtcwt_gpu.py tests/test_ext_gpu.py -q -m gpu 2>&1 | tail -1
python tools/gpu_scatj2_time.py 2>&1 | head -2
export WL_LIB=ab/libwl_segprobe.so
for cfg in dtcwt cfg5; do
for n in 0 2 3 4 6 8 12 16; do
WL_SEG_N=$n python bench.py --config $cfg --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$cfg', '$n', d['ms_per_step'], {k: r[k] for k in ('forward_ms', 'inverse_ms') if k in r} or list(r.keys())[:12])"
done; done
