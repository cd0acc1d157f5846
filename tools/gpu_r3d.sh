#!/bin/bash
OUT=gpurun_out/r03p; mkdir -p $OUT

for v in base dma; do
  if [ $v = base ]; then L=""; else L="ab/libwl_$v.so"; fi
  WL_LIB=$L PROBE=short timeout 300 python tools/gpu_strip_probe.py 2>>$OUT/err.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(json.dumps({k: d.get(k) for k in ('lib', 'case', 'max_rel_diff', 'tile_ms', 'stream_ms', 'stream_frac')}))" >> $OUT/probe.jsonl
done
cat $OUT/probe.jsonl
