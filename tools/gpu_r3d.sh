#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
WL_LIB=ab/libwl_time.so timeout 120 python tools/gpu_strip_time.py 2>>$OUT/err.log | tee -a $OUT/time.jsonl
for v in base nostage nocomp; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L PROBE=short timeout 300 python tools/gpu_strip_probe.py 2>>$OUT/err.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(json.dumps({k: d.get(k) for k in ('lib', 'case', 'max_rel_diff', 'tile_ms', 'stream_ms', 'stream_frac')}))" >> $OUT/probe.jsonl
done
cat $OUT/probe.jsonl
