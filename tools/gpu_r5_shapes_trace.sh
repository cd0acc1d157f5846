#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/r05g_shapes; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o shapes -- python $REPO/tools/gpu_r5_shapes_trace.py > $OUT/prof.log 2>&1); echo "rocprof rc=$?"
python - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/prof/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'wl_kernel' not in k: continue
    name = k.split('wl_kernel<')[-1].split('>(')[0] if 'wl_kernel<' in k else k
    acc[(name, r['Grid_Size'], r['Workgroup_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
with open(out + '/kernel_durations.csv', 'w') as g:
    g.write('kernel,grid_size,workgroup_size,launches,mean_us,min_us,max_us\n')
    for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 10: continue
        g.write('"%s",%s,%s,%d,%.2f,%.2f,%.2f\n' % (name, grid, wg, len(v), sum(v) / len(v), min(v), max(v)))
print(open(out + '/kernel_durations.csv').read())
P
