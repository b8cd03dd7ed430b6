#!/usr/bin/env python3
"""The last replayed construct() of a `rocprofv3 --kernel-trace` run of tools/build_profile.py as a timeline: start (us after the build's first
kernel), duration, queue, kernel -- what the chain of dependent launches looks like on the device.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/build_profile.py clumsydragon --k 5
    python tools/build_timeline.py gpurun_out/tl [--json out.json]
"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True)
    if not files:
        raise SystemExit('no kernel trace under ' + d)
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    first = [i for i, r in enumerate(rows) if 'k_morton_leaves' in r['Kernel_Name']]
    if not first:
        raise SystemExit('no build in the trace')
    rows = rows[first[-1]:]
    t0 = int(rows[0]['Start_Timestamp'])
    out = []
    for r in rows:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        out.append(dict(start_us=round(s / 1e3, 1), us=round((e - s) / 1e3, 1), queue=r['Queue_Id'], kernel=name, grid=int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))))
        print('%8.1f %6.1f  q%-2s %5d blocks  %s' % (out[-1]['start_us'], out[-1]['us'], r['Queue_Id'], out[-1]['grid'], name))
    end = max(o['start_us'] + o['us'] for o in out)
    busy = sum(o['us'] for o in out)
    print('%d launches, %.1f us from the first kernel\'s start to the last one\'s end, %.1f us of kernel time' % (len(out), end, busy))
    if '--json' in sys.argv:
        json.dump(dict(launches=len(out), span_us=end, kernel_us=round(busy, 1), timeline=out), open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=0)


if __name__ == '__main__':
    main()
