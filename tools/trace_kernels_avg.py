#!/usr/bin/env python3
"""average duration per kernel name of a rocprofv3 --kernel-trace directory: python tools/trace_kernels_avg.py DIR [substring]"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if sub in name:
        acc[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in acc.items():
    v = v[len(v) // 3:]                       # (the first launches include code loading)
    print('%-40s %3d launches  mean %.1f us  min %.1f' % (k, len(v), sum(v) / len(v), min(v)))
