"""Copies the judged summaries of gpurun_out/<round>/ (tools/profile_round.sh) into profiles/ and derives
profiles/traffic_latest.json (HBM bytes per traversal launch pair from the FETCH_SIZE / WRITE_SIZE passes)."""
import collections, csv, glob, json, os, shutil, sys
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
R = sys.argv[1] if len(sys.argv) > 1 else 'r01'
stats = newest("gpurun_out/%s/stats/*/*_kernel_stats.csv" % R)
shutil.copy(stats, 'profiles/%s_bench_kernel_stats.csv' % R)
rows = []
for p in ['pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_tcc']:
    fs = sorted(glob.glob("gpurun_out/%s/%s/*/*_counter_collection.csv" % (R, p)), key=os.path.getmtime)
    if not fs: continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in d:
        if k.startswith('k_'):
            for c, v in d[k].items(): rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
rows.sort()
with open('profiles/%s_bench_pmc_summary.csv' % R, 'w') as f:
    f.write('kernel,counter,dispatches,mean_per_dispatch,min,max\n')
    for r in rows: f.write('%s,%s,%d,%.3f,%.3f,%.3f\n' % r)
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for k, c, n, mean, _, _ in rows:
    tot[(k, c)] = mean * n; cnt[(k, c)] = n
# steady-state traversal kernels carry the template tag <1, 0> (calibration candidates of the first frame are <1, 1>);
# one frame launch = one first-phase kernel (k_trace or k_trace_packet) + its k_trace_long (+ k_resolve_split)
steady = [k for k in {k for k, _ in tot} if k.startswith(('k_trace', 'k_resolve_split')) and k.endswith('<1, 0>')]
frames = sum(cnt[(k, 'FETCH_SIZE')] for k in steady if k.startswith(('k_trace<', 'k_trace_packet<', 'k_trace_packet4<')))
fetch = sum(tot[(k, 'FETCH_SIZE')] for k in steady) / max(frames, 1)
write = sum(tot[(k, 'WRITE_SIZE')] for k in steady) / max(frames, 1)
json.dump({'round': R, 'source': 'profiles/%s_bench_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 4; steady-state kernels only, summed over the kernels of a frame launch and averaged over %d frame launches)' % (R, frames),
           'fetch_size_kb_raw': fetch, 'write_size_kb': write,
           'correction': 'FETCH_SIZE doubled (gfx950 reports half of a streamed read, MI355X_MICROARCH.md HBM section); uncalibrated for this access pattern',
           'hbm_bytes_per_launch': int((2 * fetch + write) * 1024)}, open('profiles/traffic_latest.json', 'w'), indent=1)
print(open('profiles/traffic_latest.json').read())
for l in open('profiles/%s_bench_kernel_stats.csv' % R).read().splitlines()[:6]: print(l)
for k, c, n, mean, _, _ in rows:
    if 'k_trace' in k: print(k, c, n, '%.4g' % mean)
# frame-launch duration from the kernel-trace stats: must agree with bench.py's live HIP-event figure (roofline.avg_launch_ms)
ks = list(csv.DictReader(open('profiles/%s_bench_kernel_stats.csv' % R)))
nm = lambda r: r['Name'].split('(')[0].replace('void ', '')
def launch_summary(tag):
    st = [r for r in ks if nm(r).startswith(('k_trace', 'k_resolve_split')) and nm(r).endswith('<1, %d>' % tag)]
    nfr = sum(int(r['Calls']) for r in st if nm(r).startswith(('k_trace<', 'k_trace_packet<', 'k_trace_packet4<')))
    return {'frame_launches': nfr, 'avg_frame_launch_ms': sum(float(r['TotalDurationNs']) for r in st) / max(nfr, 1) / 1e6}
iso, conc = launch_summary(2), launch_summary(0)
summ = {'round': R, 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline',
        'isolated_launches_tag_2': iso, 'timed_region_launches_tag_0': conc,
        'frame_launches': iso['frame_launches'], 'avg_frame_launch_ms': iso['avg_frame_launch_ms'],
        'note': 'avg_frame_launch_ms = summed duration of the <1, 2> traversal kernels / number of <1, 2> frame launches: the launches '
                'bench.py times with HIP events for roofline.avg_launch_ms (one scene at a time).  <1, 0> launches overlap on five streams.',
        'kernels': {nm(r): {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'total_ms': float(r['TotalDurationNs']) / 1e6} for r in ks if nm(r).startswith('k_')}}
json.dump(summ, open('profiles/%s_frame_launch_summary.json' % R, 'w'), indent=1)
print('avg frame launch ms (rocprof): isolated', iso, 'timed region', conc)
try:
    tl = newest("gpurun_out/%s/tlg_stats/*/*_kernel_stats.csv" % R)
    shutil.copy(tl, 'profiles/%s_tlg_config4_kernel_stats.csv' % R)
    line = [l for l in open('gpurun_out/%s/tlg_config4.log' % R) if l.startswith('{')][-1]
    open('profiles/%s_tlg_config4_line.json' % R, 'w').write(line)
    print(line)
except Exception as e:
    print('no tlg profile:', e)
try:
    c3 = newest("gpurun_out/%s/config3_stats/*/*_kernel_stats.csv" % R)
    shutil.copy(c3, 'profiles/%s_config3_kernel_stats.csv' % R)
    ls = open('gpurun_out/%s/config3.log' % R).read().splitlines()
    a = ls.index('{'); b = ls.index('}', a)
    t = '\n'.join(ls[a:b + 1]) + '\n'
    open('profiles/%s_config3_line.json' % R, 'w').write(t)
    print(t)
except Exception as e:
    print('no config3 profile:', e)
