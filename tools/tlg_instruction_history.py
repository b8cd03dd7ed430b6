"""The grid walk of config 4 over the steps of round 5's second part: instructions issued per launch by kind (rocprofv3 --pmc SQ_INSTS_*, two passes per
step: tools/round5/s2c.sh) beside the launch time measured in the same call (tools/tlg_config4.py, HIP events).  The time follows the TOTAL number of
instructions a launch issues -- scalar ones included -- at 0.95-0.96 ns per million: what the steps cut were instructions, of either kind.
   python tools/tlg_instruction_history.py > profiles/r05_tlg_instruction_counts.json
Reads gpurun_out/<tag>/ as gpurun merged them (scratch: the summary is what is committed)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = [('s2c', 'packet walk, first form: record / planes / leaf record / triangle of a cell all lanes stand in through the scalar cache'),
         ('s2e', '+ the wave\'s direction signs as scalar masks, per-camera plane table (lo - o, hi - o, origin-inside flag)'),
         ('s2h', 'experiment, not kept: the walk as a loop of the WAVE with the cell\'s indices carried in scalar registers (fewer vector, more scalar instructions)'),
         ('s2i', '+ the end of the triangle test as a chain of v_cmpx'),
         ('s2k', '+ 32-byte leaf records with six triangle ids, a leaf cell\'s entries without a loop'),
         ('s2l', '+ camera planes behind the cell records (one base pointer), near / far by load offset'),
         ('s2q', '+ the packet walk compiled once per direction octant (signs = constants of the instance)'),
         ('s2r', '+ a leaf grid of one cell on its own short path, plane offsets as loop constants'),
         ('s2s', '+ lane-by-lane leaf cells of the packet walk from the 32-byte record too'),
         ('s2t', '+ v, t and their compares behind a branch of the wave (skipped instructions were cheap ones: the time fell by 3 %, the count by 8 %)')]
KINDS = ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD')
rows = []
for tag, what in STEPS:
    d = os.path.join(ROOT, 'gpurun_out', tag)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/pmc_sq*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_tlg_trace<' in r['Kernel_Name'].split('(')[0]:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    # (one row per launch and counter: 21 launches of the tool's run, 32 since the tool also times the fused frame)
    c = {k: agg[k][1] / agg[k][0] for k in KINDS if k in agg}
    ms = [json.load(open(f))['trace_ms'] for f in sorted(glob.glob(d + '/tlg_*.json'))]
    if not c or not ms:
        continue
    total = sum(c.values())
    t = sum(ms) / len(ms)
    rows.append(dict(step=tag, what=what, trace_ms=round(t, 3), instructions_per_launch_M={k[9:]: round(v / 1e6, 1) for k, v in c.items()}, total_M=round(total / 1e6, 1),
                     ns_per_million_instructions=round(t * 1e6 / (total / 1e6) / 1e3, 4), wave_time_parked_on_waitcnt=(round(agg['SQ_WAIT_ANY'][1] / agg['SQ_WAVE_CYCLES'][1], 3) if agg.get('SQ_WAVE_CYCLES') else None)))
json.dump(dict(what=__doc__.split('\n   python')[0], round4=dict(trace_ms=3.33, VALU_M=2282.0), round5_first_part=dict(trace_ms=2.289, VALU_M=1511.4, SALU_M=618.0, SMEM_M=0.3,
               note='vector-memory and LDS instructions of that build were not counted'), steps=rows), sys.stdout, indent=1)
print()
