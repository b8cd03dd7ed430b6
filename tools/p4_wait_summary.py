#!/usr/bin/env python3
"""Summarises the counter passes of tools/p4_wait_pmc.sh: per traversal kernel tagged <1, 2>, counter sums per dispatch."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    res = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True):
        tag = os.path.relpath(f, out).split(os.sep)[0]
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            if not k.startswith(('k_trace', 'k_resolve')) or ', 2' not in k.split('<', 1)[-1][:8]:
                continue
            res[k][row['Counter_Name']] += float(row['Counter_Value'])
            disp[k][tag].add(row['Dispatch_Id'])
    summary = {}
    for k, c in res.items():
        n = max(len(v) for v in disp[k].values())
        per = {name: v / n for name, v in c.items()}
        d = dict(dispatches=n, per_dispatch={a: round(b, 1) for a, b in sorted(per.items())})
        g = per.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        if g:
            d['cycles_per_dispatch'] = round(g, 1)
            d['valu_issue_frac'] = round(per.get('SQ_INSTS_VALU', 0) * 2.0 / (1024 * g), 4)
        if per.get('SQC_DCACHE_REQ'):
            d['scalar_cache_hit_rate'] = round(per.get('SQC_DCACHE_HITS', 0) / per['SQC_DCACHE_REQ'], 4)
        if per.get('SQ_WAVE_CYCLES'):
            d['wait_inst_any_over_wave_cycles'] = round(per.get('SQ_WAIT_INST_ANY', 0) / per['SQ_WAVE_CYCLES'], 4)
        summary[k] = d
    passes = open(os.path.join(out, 'passes.txt')).read().splitlines() if os.path.exists(os.path.join(out, 'passes.txt')) else []
    print(json.dumps(dict(passes=passes, kernels=summary), indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
