#!/usr/bin/env python3
"""The last replayed construct() of a `rocprofv3 --kernel-trace` run of tools/build_profile.py as a timeline: start (us after the build's first
kernel), duration, queue, kernel -- what the chain of dependent launches looks like on the device.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/build_profile.py clumsydragon --k 5
    python tools/build_timeline.py gpurun_out/tl [--n TRIANGLES] [--json out.json]
(--n: + the stages against SURVEY.md 8(d)'s build-stage bytes)
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
    stages = None
    if '--n' in sys.argv:
        # SURVEY.md 8(d), build stages, ALGORITHMIC bytes (layout-independent), here for the two hierarchies a build produces:
        #   Morton n (36 in + 8 + 48 out); sort N x 8 B x 2 per pass (bitonic: log2 N (log2 N + 1) / 2 passes over N = the next power of two;
        #   radix: 4 passes over n); tree (n - 1) (6 x 8 B reads + 48 B write); refit (2n - 1) x 48 x 2; the records behind the refit
        #   (64-byte traversal records of both trees, the search tree's 128-byte four-wide records, 40-byte triangle records) are this library's own
        n = int(sys.argv[sys.argv.index('--n') + 1])
        N = 1 << (n - 1).bit_length()
        lg = N.bit_length() - 1
        trees = 2
        radix = any('k_radix_hist' in o['kernel'] or 'k_radix_small' in o['kernel'] for o in out)
        network = any('bitonic' in o['kernel'] for o in out)
        # (FAST: both pair arrays through the radix sort; FIXED: both through the network up to 2^18 pairs, beyond that the search tree's through the radix sort)
        sort_bytes = (N * 16 * lg * (lg + 1) // 2) * (trees if network and not radix else (1 if network else 0)) + (n * 16 * 4) * (trees if radix and not network else (1 if radix else 0))
        algo = {
            'morton + leaves': n * (36 + (8 + 48) * trees),
            'sort': sort_bytes,
            'radix tree': (n - 1) * (48 + 48) * trees,
            'refit': (2 * n - 1) * 48 * 2 * trees,
            'records (finish)': (n - 1) * 64 * trees + (n - 1) * 128 + n * (40 + 4 * trees),
        }
        def stage_of(k):
            if 'morton' in k: return 'morton + leaves'
            if 'bitonic' in k or 'k_radix_hist' in k or 'k_radix_rows' in k or 'k_radix_scatter' in k or 'k_radix_small' in k: return 'sort'
            if 'k_radix_tree' in k: return 'radix tree'
            if 'k_seg' in k or 'k_refit' in k or 'k_rq' in k or 'k_scan' in k: return 'refit'
            if 'k_finish' in k: return 'records (finish)'
            return 'other'
        us = {}
        launches = {}
        for o in out:
            st = stage_of(o['kernel']); us[st] = us.get(st, 0.0) + o['us']; launches[st] = launches.get(st, 0) + 1
        stages = []
        print('stage                launches      us   algorithmic MB   GB/s   of 8 TB/s')
        for st in ('morton + leaves', 'sort', 'radix tree', 'refit', 'records (finish)', 'other'):
            if st not in us:
                continue
            b = algo.get(st, 0)
            gbps = b / (us[st] * 1e-6) / 1e9 if us[st] else 0.0
            stages.append(dict(stage=st, launches=launches[st], us=round(us[st], 1), algorithmic_bytes=b, gbps=round(gbps, 1), frac_of_hbm_peak=round(gbps / 8000.0, 4)))
            print('%-20s %8d %7.1f %16.2f %6.0f %10.3f' % (st, launches[st], us[st], b / 1e6, gbps, gbps / 8000.0))
    end = max(o['start_us'] + o['us'] for o in out)
    busy = sum(o['us'] for o in out)
    print('%d launches, %.1f us from the first kernel\'s start to the last one\'s end, %.1f us of kernel time' % (len(out), end, busy))
    if '--json' in sys.argv:
        json.dump(dict(launches=len(out), span_us=end, kernel_us=round(busy, 1), stages=stages, timeline=out), open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=0)


if __name__ == '__main__':
    main()
