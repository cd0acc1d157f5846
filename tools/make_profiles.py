"""Turn the rocprofv3 outputs of one gpurun call into the tracked summaries under profiles/:
   <tag>_kernel_durations.csv : per (kernel, grid size) launch count / mean / min / max duration (us), from the kernel trace
   <tag>_hbm_traffic.json     : per kernel FETCH_SIZE / WRITE_SIZE (PMC passes) with the gfx950 correction of
                                MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced reads) and the
                                digest of the sources the numbers were measured on (bench.py quotes them only on a match)
usage: python tools/make_profiles.py <tag> <kernel_trace.csv> [<pmc_summary.json>]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, trace = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    name = r['Kernel_Name']
    if 'wl_kernel' not in name:
        continue
    short = name.replace('void ', '').split('(')[0].replace('wl_kernel<', '')[:-1]
    grid = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
    rows[(short, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = os.path.join(ROOT, 'profiles', tag + '_kernel_durations.csv')
with open(out, 'w') as f:
    f.write('kernel,grid_size,launches,mean_us,min_us,max_us\n')
    for (k, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%d,%.2f,%.2f,%.2f\n' % (k, g, len(v), sum(v) / len(v), min(v), max(v)))
print('wrote', out)
if len(sys.argv) > 3:
    import bench
    pmc = json.load(open(sys.argv[3]))
    kernels = {}
    for k, c in pmc.items():
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            short = k.replace('wl_kernel<', '')[:-1] if k.startswith('wl_kernel<') else k
            fs, ws = c['FETCH_SIZE']['max'] * 1024, c['WRITE_SIZE']['max'] * 1024   # counters are in KB
            kernels[short] = {'fetch_bytes_raw': fs, 'write_bytes_raw': ws, 'hbm_bytes_corrected': 2 * fs + ws,
                              'note': 'largest dispatch of the pass; FETCH_SIZE doubled (gfx950: half the bytes of wide '
                                      'coalesced reads are counted), WRITE_SIZE as is'}
    tj = {'source_digest': bench.source_digest(), 'kernels': kernels}
    out = os.path.join(ROOT, 'profiles', tag + '_hbm_traffic.json')
    json.dump(tj, open(out, 'w'), indent=1)
    print('wrote', out, json.dumps(kernels)[:400])
