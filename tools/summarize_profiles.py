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
m = {(k, c): mean for k, c, n, mean, _, _ in rows}
fetch = m.get(('k_trace<1>', 'FETCH_SIZE'), 0) + m.get(('k_trace_long<1>', 'FETCH_SIZE'), 0)
write = m.get(('k_trace<1>', 'WRITE_SIZE'), 0) + m.get(('k_trace_long<1>', 'WRITE_SIZE'), 0)
json.dump({'round': R, 'source': 'profiles/%s_bench_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 4; includes the calibration launches of the first frame)' % R,
           'fetch_size_kb_raw': fetch, 'write_size_kb': write,
           'correction': 'FETCH_SIZE doubled (gfx950 reports half of a streamed read, MI355X_MICROARCH.md HBM section); uncalibrated for this access pattern',
           'hbm_bytes_per_launch': int((2 * fetch + write) * 1024)}, open('profiles/traffic_latest.json', 'w'), indent=1)
print(open('profiles/traffic_latest.json').read())
for l in open('profiles/%s_bench_kernel_stats.csv' % R).read().splitlines()[:6]: print(l)
for k, c, n, mean, _, _ in rows:
    if 'k_trace' in k: print(k, c, n, '%.4g' % mean)
