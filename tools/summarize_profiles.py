"""Copies the judged summaries of gpurun_out/<round>/ (tools/profile_round.sh) into profiles/."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as bench_mod
newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)
R = sys.argv[1] if len(sys.argv) > 1 else 'r05'
src = 'gpurun_out/%s' % R
stats = newest(src + '/stats/**/*_kernel_stats.csv')
shutil.copy(stats, 'profiles/%s_bench_kernel_stats.csv' % R)
line = [l for l in open(src + '/bench_line.json') if l.startswith('{')][-1]
open('profiles/%s_bench_line.json' % R, 'w').write(line)
shutil.copy(src + '/bench_details.json', 'profiles/%s_bench_details.json' % R)      # the whole record the line summarises (bench.py --details)
bench = json.load(open(src + '/bench_details.json'))
# the counter rows the bench line was computed from (bench.py --keep-pmc), summed per kernel and counter
rows = []
for ln in open(src + '/pmc/bench_pmc_rows.csv').read().splitlines()[1:]:
    head, counter, value = ln.rsplit(',', 2)                     # kernel names hold commas (k_trace<1, 2>)
    pas, disp, kernel = head.split(',', 2)
    rows.append(dict(kernel=kernel.strip('"'), counter=counter, value=value))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[(r['kernel'], r['counter'])]; a[0] += 1; a[1] += float(r['value'])
with open('profiles/%s_bench_pmc_summary.csv' % R, 'w') as f:
    f.write('kernel,counter,dispatches,mean_per_dispatch\n')
    for (k, c), (n, v) in sorted(agg.items()):
        f.write('"%s",%s,%d,%.3f\n' % (k, c, n, v / n))
# frame-launch duration from the kernel-trace stats: must agree with bench.py's live HIP-event figure (roofline.avg_launch_ms)
ks = list(csv.DictReader(open('profiles/%s_bench_kernel_stats.csv' % R)))
nm = lambda r: r['Name'].split('(')[0].replace('void ', '')
def launch_summary(tag):
    st = [r for r in ks if nm(r).startswith(('k_trace', 'k_resolve_split')) and bench_mod.kernel_tag(nm(r)) == (1, tag)]
    nfr = sum(int(r['Calls']) for r in st if nm(r).startswith(('k_trace<', 'k_trace_packet<', 'k_trace_packet4<')))
    return {'frame_launches': nfr, 'avg_frame_launch_ms': sum(float(r['TotalDurationNs']) for r in st) / max(nfr, 1) / 1e6}
iso, conc = launch_summary(2), launch_summary(0)
summ = {'round': R, 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc',
        'isolated_launches_tag_2': iso, 'timed_region_launches_tag_0': conc,
        'bench_line_avg_launch_ms': bench['roofline']['avg_launch_ms'],
        'note': 'avg_frame_launch_ms of the <1, 2> kernels = the launches bench.py times with HIP events for roofline.avg_launch_ms (one scene at a '
                'time); <1, 0> launches overlap on five streams; <1, 1> = first-frame calibration',
        'kernels': {nm(r): {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'total_ms': float(r['TotalDurationNs']) / 1e6} for r in ks if nm(r).startswith('k_')}}
json.dump(summ, open('profiles/%s_frame_launch_summary.json' % R, 'w'), indent=1)
print('avg frame launch ms: rocprof isolated', iso, 'timed region', conc, 'bench line', bench['roofline']['avg_launch_ms'])
for name, sub, log in (('tlg_config4', 'tlg_stats', 'tlg_config4.log'), ('config3', 'config3_stats', 'config3.log')):
    try:
        shutil.copy(newest(src + '/%s/**/*_kernel_stats.csv' % sub), 'profiles/%s_%s_kernel_stats.csv' % (R, name))
        txt = open(src + '/' + log).read()
        a = txt.index('{'); b = txt.rindex('}')
        open('profiles/%s_%s_line.json' % (R, name), 'w').write(txt[a:b + 1] + '\n')
    except Exception as e:
        print('no %s profile: %s' % (name, e))
# counter passes of configs 3 and 4 (tools/profile_round.sh): measured HBM bytes and VALU issue of the steady-state trace launches,
# folded into the _line.json of the configuration (FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md)
def counter_pass(prefix, kernel_prefixes, skip_first):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in ('fetch', 'write', 'sq'):
        try:
            f = newest(src + '/%s_%s/**/*counter_collection.csv' % (prefix, tag))
        except ValueError:
            return None
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            if k.startswith(kernel_prefixes) and (not k.startswith(('k_trace', 'k_resolve')) or bench_mod.kernel_tag(k) == (1, 0)):   # (<1, 1> = calibration)
                per[k][row['Counter_Name']].append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
    out = {}
    for k, cs in per.items():
        d = {}
        for c, vals in cs.items():
            by = collections.defaultdict(float)
            for disp, v in vals:
                by[disp] += v
            steady = [by[x] for x in sorted(by)][skip_first:] or [by[x] for x in sorted(by)]     # (the first launches calibrate / warm up)
            d[c] = sum(steady) / len(steady)
        out[k] = d
    return out
for name, prefix, kp, skip in (('config3', 'config3_pmc', ('k_trace', 'k_resolve'), 1), ('tlg_config4', 'tlg_pmc', ('k_tlg_trace<',), 1)):
    try:
        pk = counter_pass(prefix, kp, skip)
        if not pk:
            raise RuntimeError('no counter files')
        line_path = 'profiles/%s_%s_line.json' % (R, name)
        line = json.load(open(line_path))
        # the kernel with the most device time carries the launch
        dom = max(pk, key=lambda k: pk[k].get('GRBM_GUI_ACTIVE', 0.0))
        c = pk[dom]
        cycles = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        hbm = sum((2.0 * v.get('FETCH_SIZE', 0.0) + v.get('WRITE_SIZE', 0.0)) * 1024 for v in pk.values())
        ms = line.get('trace_ms') or line.get('ms_per_frame') or next((v['ms_per_frame'] for v in line.values() if isinstance(v, dict) and 'ms_per_frame' in v), 0.0)
        line['counters'] = dict(source='rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, SQ group: one pass each), steady-state launches',
                                dominant_kernel=dom, per_launch={k: {a: round(b, 1) for a, b in v.items()} for k, v in pk.items()},
                                hbm_bytes_per_launch=int(hbm), measured_hbm_gbps=(round(hbm / (ms * 1e-3) / 1e9, 1) if ms else None),
                                hbm_frac_of_8tbps=(round(hbm / (ms * 1e-3) / 1e9 / 8000.0, 4) if ms else None),
                                valu_issue_frac=(round(c.get('SQ_INSTS_VALU', 0.0) * 2.0 / (1024 * cycles), 4) if cycles else None),
                                wave_time_split=(dict(executing=round(c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES'], 3), parked_on_waitcnt=round(c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'], 3),
                                                      issue_stall=round(c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES'], 3)) if c.get('SQ_WAVE_CYCLES') else None))
        if name == 'tlg_config4' and line.get('per_ray', {}).get('top_cells') and c.get('SQ_WAVES'):
            # vector instructions a wave issues per top-level cell one of its rays walks (all loop levels below it included): what the verdict's
            # "instructions per top cell" asks for; round 3 (profiles/r03_tlg_config4_line.json): 2 282 M / 32 640 waves / 152.2 cells = 459
            # (waves of a launch = rays / 64: SQ_WAVES is sampled per shader engine and reads low on some passes of round 5)
            waves = 1920 * 1080 / 64.0
            line['counters']['valu_instructions_per_wave'] = round(c.get('SQ_INSTS_VALU', 0.0) / waves, 1)
            line['counters']['valu_instructions_per_top_cell'] = round(c.get('SQ_INSTS_VALU', 0.0) / waves / line['per_ray']['top_cells'], 1)
            line['counters']['valu_instructions_per_top_cell_round3'] = 459.4
            line['counters']['floor_ms_at_full_issue'] = round(c.get('SQ_INSTS_VALU', 0.0) * 2.0 / 1024.0 / (cycles / ms) , 3) if cycles and ms else None
        json.dump(line, open(line_path, 'w'), indent=1)
        print(name, 'counters:', json.dumps({k: line['counters'][k] for k in ('dominant_kernel', 'hbm_bytes_per_launch', 'measured_hbm_gbps', 'hbm_frac_of_8tbps', 'valu_issue_frac', 'wave_time_split')}))
    except Exception as e:
        print('no %s counters: %s' % (name, e))
# the grid walk's vector instructions by class (one more counter pass), priced with the issue cost of each class at full occupancy
# (tools/micro/pk_rate.hip, ns per wave-instruction per SIMD with eight waves per SIMD): what the launch costs in issue slots alone
try:
    mix = collections.defaultdict(lambda: [0, 0.0])
    for tag in ('mix', 'cyc'):
        f = newest(src + '/tlg_pmc_%s/**/*counter_collection.csv' % tag)
        by = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].split('(')[0].replace('void ', '').strip().startswith('k_tlg_trace<'):      # (not k_tlg_trace_records, the build's last launch)
                by[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
        for c, d in by.items():
            vals = [d[x] for x in sorted(d)][1:] or list(d.values())
            mix[c] = sum(vals) / len(vals)
    cost = dict(add=1.2, mul=1.2, fma=1.6, trans=3.4, int32=1.5, cvt=1.75, other=1.8)         # ns per wave-instruction per SIMD (r05_valu_issue_costs.txt)
    cnt = dict(add=mix.get('SQ_INSTS_VALU_ADD_F32', 0.0), mul=mix.get('SQ_INSTS_VALU_MUL_F32', 0.0), fma=mix.get('SQ_INSTS_VALU_FMA_F32', 0.0),
               trans=mix.get('SQ_INSTS_VALU_TRANS_F32', 0.0), int32=mix.get('SQ_INSTS_VALU_INT32', 0.0), cvt=mix.get('SQ_INSTS_VALU_CVT', 0.0))
    cnt['other'] = max(mix.get('SQ_INSTS_VALU', 0.0) - sum(cnt.values()), 0.0)
    line_path = 'profiles/%s_tlg_config4_line.json' % R
    line = json.load(open(line_path))
    priced_ms = sum(cnt[k] * cost[k] for k in cnt) / 1024.0 * 1e-6
    floor_ms = (sum(cnt.values()) + cnt['trans'] * (cost['trans'] / cost['add'] - 1.0)) * cost['add'] / 1024.0 * 1e-6
    line['valu_issue_roof'] = dict(what='vector instructions per launch by class (rocprofv3 --pmc SQ_INSTS_VALU_*) against the issue costs of tools/micro/pk_rate.hip (ns per '
                                        'wave-instruction per SIMD with all wave slots busy).  floor_ms: EVERY instruction at the rate of the cheapest class (v_add / v_mul / v_mov: two '
                                        'of different waves issue side by side), transcendentals at theirs -- no instruction stream of this length runs faster.  priced_by_class_ms: every '
                                        'class at the cost it shows alone; it exceeds the measured time (classes overlap in the two pipes, and "other" / int32 hold cheap moves and adds), '
                                        'so it is an estimate, not a bound',
                                   counts={k: round(v) for k, v in cnt.items()}, cost_ns_per_wave_instruction_per_simd=cost,
                                   floor_ms=round(floor_ms, 3), priced_by_class_ms=round(priced_ms, 3), trace_ms=line.get('trace_ms'),
                                   frac=(round(floor_ms / line['trace_ms'], 3) if line.get('trace_ms') else None),
                                   cycles_per_instruction_per_simd=(round(mix['GRBM_GUI_ACTIVE'] / 8.0 / (sum(cnt.values()) / 1024.0), 2) if mix.get('GRBM_GUI_ACTIVE') else None),
                                   raw={k: round(v, 1) for k, v in mix.items()})
    json.dump(line, open(line_path, 'w'), indent=1)
    print('tlg issue floor', line['valu_issue_roof']['floor_ms'], 'priced', line['valu_issue_roof']['priced_by_class_ms'], 'of', line.get('trace_ms'))
except Exception as e:
    print('no instruction mix of the grid walk: %s' % e)
# every instruction the grid walk issues, by kind: the launch time of round 5's builds followed their total at 0.95-0.96 ns per million
try:
    f = newest(src + '/tlg_pmc_kinds/**/*counter_collection.csv')
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name'].split('(')[0].replace('void ', '').strip().startswith('k_tlg_trace<'):
            by[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
    kinds = {}
    for c, d in by.items():
        vals = [d[x] for x in sorted(d)][1:] or list(d.values())
        kinds[c] = sum(vals) / len(vals)
    line_path = 'profiles/%s_tlg_config4_line.json' % R
    line = json.load(open(line_path))
    issued = sum(v for k, v in kinds.items() if k.startswith('SQ_INSTS_') and k != 'SQ_INSTS_BRANCH')
    line['instructions_issued'] = dict(what='instructions of every kind per launch (rocprofv3 --pmc SQ_INSTS_*; branches are among the scalar ones).  Through the steps of round 5 the launch '
                                            'time followed their TOTAL at 0.95-0.96 ns per million (profiles/r05_tlg_instruction_counts.json): the walk is bound by the rate at which its seven '
                                            'waves per SIMD issue instructions of any kind, each wave one at a time',
                                       per_launch_M={k[9:]: round(v / 1e6, 1) for k, v in kinds.items() if k.startswith('SQ_INSTS_')}, total_M=round(issued / 1e6, 1),
                                       per_top_level_cell=round(issued / (1920 * 1080 / 64.0) / line['per_ray']['top_cells'], 1),
                                       ns_per_million=(round(line['trace_ms'] * 1e6 / (issued / 1e6) / 1e3, 4) if line.get('trace_ms') else None))
    json.dump(line, open(line_path, 'w'), indent=1)
    print('tlg instructions issued', line['instructions_issued']['per_launch_M'], line['instructions_issued']['ns_per_million'])
except Exception as e:
    print('no instruction kinds of the grid walk: %s' % e)
for f in ('pk_rate.txt', 'div_check.txt'):
    try:
        if not os.path.exists(src + '/' + f) or 'No such file' in open(src + '/' + f).read(2000):
            raise RuntimeError('not run in this round (the record in profiles/ is kept)')
        shutil.copy(src + '/' + f, 'profiles/%s_%s' % (R, {'pk_rate.txt': 'valu_issue_costs.txt', 'div_check.txt': 'division_sequences_check.txt'}[f]))
    except Exception as e:
        print('no %s: %s' % (f, e))
for f, dst in (('config_report.json', '%s_config_report.json' % R), ('bench_group4.json', '%s_bench_group4_rehearsal.json' % R)):
    try:
        txt = open(src + '/' + f).read()
        open('profiles/' + dst, 'w').write(txt[txt.index('{'):])
    except Exception as e:
        print('no %s: %s' % (f, e))
# the ray-buffer entry point (tools/rays_profile.py: bench.py's three legs per scene, verified) and construct() per kernel (tools/build_profile.py)
for name, sub, log in (('raybuffer', 'raybuffer_stats', 'raybuffer.log'), ('build', 'build_stats', 'build.log')):
    try:
        shutil.copy(newest(src + '/%s/**/*_kernel_stats.csv' % sub), 'profiles/%s_%s_kernel_stats.csv' % (R, name))
        txt = [l for l in open(src + '/' + log).read().splitlines() if l.startswith('{')][-1]
        open('profiles/%s_%s_line.json' % (R, name), 'w').write(txt + '\n')
    except Exception as e:
        print('no %s profile: %s' % (name, e))
# construct(): the chain of launches stage by stage (tools/build_timeline.py), host-timed builds of every scene in both modes, the dynamic-scene frame
try:
    tl = {}
    for scene in ('dragon', 'atrium'):
        for m in (0, 2):
            f = src + '/build_timeline_%s_mode%d.json' % (scene, m)
            if os.path.exists(f):
                tl['%s, %s' % (scene, 'FIXED' if m == 0 else 'FAST')] = json.load(open(f))
    times = {}
    for tag in ('fixed', 'fast', 'heightfield_fixed', 'heightfield_fast'):
        f = src + '/build_times_%s.json' % tag
        if os.path.exists(f):
            for k, v in json.load(open(f)).items():
                times.setdefault(k, {})['FAST' if tag.endswith('fast') else 'FIXED'] = v
    dyn = [l.strip() for l in open(src + '/dynamic_frames.txt')] if os.path.exists(src + '/dynamic_frames.txt') else []
    json.dump(dict(what='construct() at the end of the round: host-timed builds (initializeFrame + construct, replayed graph), the last build of a kernel trace '
                        'stage by stage against SURVEY.md 8(d) build-stage bytes (both hierarchies), and the dynamic-scene frame (tools/dyn_profile.py, five frames '
                        'in flight policy as in bench.py)', build_ms=times, dynamic_frames=dyn, timelines=tl), open('profiles/%s_build_timeline.json' % R, 'w'), indent=0)
    for k, v in times.items():
        print('build', k, {m: x['build_ms'] for m, x in v.items()})
except Exception as e:
    print('no build timeline: %s' % e)
try:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for tag in ('fetch', 'write', 'sq'):
        f = newest(src + '/raybuffer_pmc_%s/**/*counter_collection.csv' % tag)
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            if k.startswith(('k_trace', 'k_resolve')) and bench_mod.kernel_tag(k) in ((0, 0), (0, 2)):      # ray-buffer instances, steady state (<0, 1> = calibration)
                a = agg[(k, row['Counter_Name'])]; a[0] += 1; a[1] += float(row['Counter_Value'])
    with open('profiles/%s_raybuffer_pmc_summary.csv' % R, 'w') as f:
        f.write('kernel,counter,samples,mean_per_sample\n')
        for (k, c), (n, v) in sorted(agg.items()):
            f.write('"%s",%s,%d,%.3f\n' % (k, c, n, v / n))
    ks2 = {}
    for (k, c), (n, v) in agg.items():
        ks2.setdefault(k, {})[c] = v / n
    for k, c in ks2.items():
        cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        print('ray buffer', k, 'VALU issue', round(c.get('SQ_INSTS_VALU', 0) * 2.0 / (1024 * cyc), 4) if cyc else None,
              'HBM bytes per dispatch', int((2.0 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024))
except Exception as e:
    print('no ray-buffer counters: %s' % e)
try:
    shutil.copy(src + '/raybuffer_sweep.json', 'profiles/%s_raybuffer_sweep.json' % R)
except Exception as e:
    print('no ray-buffer sweep: %s' % e)
print(json.dumps({k: bench[k] for k in ('value', 'ms_per_step', 'roofline', 'roofline_valu_issue')}, indent=1)[:3000])
