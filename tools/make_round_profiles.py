"""Turn the outputs of tools/gpu_round5.sh / gpu_round4.sh / gpu_round3.sh (one gpurun call) into the tracked summaries under profiles/
(RND = the round prefix, 'r04' unless given; the counter files are written only when the call made the PMC passes):
   <tag>_bench_line.json, <tag>_bench_{dtcwt,scat,cfg5}.json   the JSON lines of the four bench commands
   <tag>_kernel_durations.csv, <tag>_{dtcwt,scat,cfg5}_kernel_durations.csv   per (kernel, grid) launch count / mean / min / max (us)
   RND_hbm_traffic.json        FETCH_SIZE / WRITE_SIZE of the metric's kernels (gfx950 correction) + the digest of the sources
                               (bench.py quotes it only on a match)
   <tag>_{dtcwt,scat,cfg5}_hbm_traffic.json   the same for the other configs
   RND_cfg5_pmc_summary.json   SQ counters of the two config-5 kernels and what they say about the bound
usage: python tools/make_round_profiles.py <tag> [<round prefix>]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else 'r04'
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def short(name):
    name = name.replace('void ', '').split('(')[0]
    return name.replace('wl_kernel<', '')[:-1].strip() if name.startswith('wl_kernel<') else name.strip()


_DM = {}


def demangle(k):
    import subprocess
    if k in _DM:
        return _DM[k]
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            _DM[k] = subprocess.run([tool, k], stdout=subprocess.PIPE, check=True).stdout.decode().strip()
            return _DM[k]
        except Exception:
            continue
    return k


def pretty(k):
    """c++filt does not know _Float16 (DF16_): our kernels are wl_kernel<Functor<T, ints...>>, which is easy to read back."""
    import re
    d = demangle(k)
    if not d.startswith('_Z'):
        return d
    m = re.match(r'_Z9wl_kernelI(\d+)(.*)EvNT_4ArgsE$', k)
    if not m:
        return k
    body = m.group(2)
    name, rest = body[:int(m.group(1))], body[int(m.group(1)):]
    args = []
    if rest.startswith('I'):
        rest = rest[1:]
        while rest and rest[0] != 'E':
            if rest.startswith('DF16_'):
                args.append('_Float16'); rest = rest[5:]
            elif rest[0] == 'f':
                args.append('float'); rest = rest[1:]
            elif rest[0] == 'd':
                args.append('double'); rest = rest[1:]
            else:
                mm = re.match(r'L[ib](n?\d+)E', rest)
                if not mm:
                    return k
                args.append(mm.group(1).replace('n', '-')); rest = rest[mm.end():]
    return 'wl_kernel<%s<%s>>' % (name, ', '.join(args))


def durations(trace, out):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if 'wl_kernel' not in r['Kernel_Name']:
            continue
        grid = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
        rows[(short(pretty(r['Kernel_Name']) if r['Kernel_Name'].startswith('_Z') else r['Kernel_Name']), grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    with open(out, 'w') as f:
        f.write('kernel,grid_size,launches,mean_us,min_us,max_us\n')
        for (k, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.2f,%.2f,%.2f\n' % (k, g, len(v), sum(v) / len(v), min(v), max(v)))
    print('wrote', out)


def traffic(pmc, out):
    kernels = {}
    for k, c in json.load(open(pmc)).items():
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            name = pretty(k)
            fs, ws = c['FETCH_SIZE']['max'] * 1024, c['WRITE_SIZE']['max'] * 1024   # counters are in KB
            kernels[short(name)] = {'fetch_bytes_raw': fs, 'write_bytes_raw': ws, 'hbm_bytes_corrected': 2 * fs + ws,
                                    'note': 'largest dispatch of the pass; FETCH_SIZE doubled (gfx950: half the bytes of wide '
                                            'coalesced reads are counted, MI355X_MICROARCH.md), WRITE_SIZE as is'}
    json.dump({'source_digest': bench.source_digest(), 'kernels': kernels}, open(out, 'w'), indent=1)
    print('wrote', out)
    return kernels


ONLY_TRAFFIC = len(sys.argv) > 3 and sys.argv[3] == 'traffic'    # on the GPU box, between the counter passes and the bench lines
for src, dst in () if ONLY_TRAFFIC else (('bench_line.json', 'bench_line.json'), ('bench_line_20.json', 'bench_line_20.json'), ('bench_dtcwt.json', 'bench_dtcwt.json'), ('bench_scat.json', 'bench_scat.json'),
                 ('bench_cfg5.json', 'bench_cfg5.json'), ('box.txt', 'box.txt')):
    p = os.path.join(G, tag, src)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith('{')] if src.endswith('.json') else None
        with open(os.path.join(P, tag + '_' + dst), 'w') as f:
            f.write(lines[-1] if lines else open(p).read())
if not ONLY_TRAFFIC:
    durations(os.path.join(G, tag, 'prof', 'bench_kernel_trace.csv'), os.path.join(P, tag + '_kernel_durations.csv'))
for c in () if ONLY_TRAFFIC else ('dtcwt', 'scat', 'cfg5'):
    durations(os.path.join(G, tag, 'prof_' + c, 'bench_kernel_trace.csv'), os.path.join(P, '%s_%s_kernel_durations.csv' % (tag, c)))
if os.path.exists(os.path.join(G, 'pmc_%s_bench' % tag, 'pmc_summary.json')):
    traffic(os.path.join(G, 'pmc_%s_bench' % tag, 'pmc_summary.json'), os.path.join(P, RND + '_hbm_traffic.json'))
    for c in ('dtcwt', 'scat', 'cfg5'):
        t = traffic(os.path.join(G, 'pmc_%s_%s' % (tag, c), 'pmc_summary.json'), os.path.join(P, '%s_%s_hbm_traffic.json' % (RND, c)))
    # config 5: what the SQ counters say about the bound of the two strip kernels (level-1 dispatch = the largest)
    pmc = json.load(open(os.path.join(G, 'pmc_%s_cfg5' % tag, 'pmc_summary.json')))
    out = {'source_digest': bench.source_digest(), 'tag': tag,
           'how': 'rocprofv3 --pmc passes (counters + kernel trace only) around `python bench.py --config cfg5`; values of the largest '
                  'dispatch of each kernel (level 1: 32x16x2048x2048 float16), summed over the chip by rocprofv3', 'kernels': {}}
    for k, c in pmc.items():
        if 'Strip' not in k or 'SQ_INSTS_VALU' not in c:
            continue
        name = short(pretty(k))
        m = {n: c[n]['max'] for n in c}
        waves, valu = m['SQ_WAVES'], m['SQ_INSTS_VALU']
        cycles = m['GRBM_GUI_ACTIVE'] / 8.0            # the counter is summed over the 8 XCDs
        d = {'counters_level1_dispatch': m,
             'kernel_cycles': round(cycles),
             'valu_instructions_per_wave': round(valu / waves, 1),
             # a wave64 VALU instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs
             'valu_pipe_utilisation': round(4 * valu / (1024 * cycles), 4),
             'lds_bank_conflict_fraction_of_lds_active': round(m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1), 4),
             'salu_per_valu': round(m['SQ_INSTS_SALU'] / valu, 3), 'lds_per_valu': round(m['SQ_INSTS_LDS'] / valu, 3),
             'wait_inst_any_fraction_of_wave_cycles': round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 4),   # (both in units of 4 cycles)
             'hbm_bytes_corrected': t.get(name, {}).get('hbm_bytes_corrected')}
        out['kernels'][name] = d
    out['reading'] = ('A wave64 VALU instruction occupies its SIMD for 4 cycles, so 4 x SQ_INSTS_VALU / (SIMDs x kernel cycles) is the VALU-pipe '
                      'utilisation.  Rounds 3-4 (direct form, 17 packed FMAs per output sample): both kernels bound by vector-ALU issue (utilisation '
                      '0.63-0.67, VALU roofline of the four levels 1.24 ms against 1.07 ms at the HBM peak).  Round 5 (lattice column pass, '
                      'csrc/wl_lattice.h: 12.5 per output sample, VALU floor 0.91 ms): see DESIGN.md 4.18 / 5.')
    json.dump(out, open(os.path.join(P, RND + '_cfg5_pmc_summary.json'), 'w'), indent=1)
    print('wrote', RND + '_cfg5_pmc_summary.json')
