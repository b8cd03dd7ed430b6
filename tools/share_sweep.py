#!/usr/bin/env python3
"""Isolated frame time (first-phase kernel + follow-up, HIP events) of a bundled scene at 1920x1080 under forced strategies:
four-wide packet kernel with sub-trees shared after B steps, whole blocks, lane-per-ray budgets.  A new context per setting
(the knobs are read when a context is created).   python tools/share_sweep.py man tenryuu clumsydragon"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openclraytracer_amd import capi

SETTINGS = [('packet4 whole blocks', '4', 1 << 20), ('packet4 share 1024', '4', 1024), ('packet4 share 256', '4', 256), ('packet4 share 128', '4', 128),
            ('packet4 share 64', '4', 64), ('packet4 share 32', '4', 32), ('packet4 share 16', '4', 16), ('packet4 share 8', '4', 8), ('lane-per-ray 16', '0', 16), ('lane-per-ray 64', '0', 64)]


def main():
    global SETTINGS
    if os.environ.get('SWEEP') == 'lane':
        SETTINGS = [('lane-per-ray %d' % b, '0', b) for b in (6, 8, 12, 16, 24, 32, 48)]
    if os.environ.get('SWEEP') == 'small':
        SETTINGS = [x for x in SETTINGS if x[0] in ('packet4 share 64', 'packet4 share 32', 'packet4 share 16', 'packet4 share 8')]
    if os.environ.get('SWEEP') == 'quick':
        SETTINGS = [x for x in SETTINGS if x[0] in ('packet4 whole blocks', 'packet4 share 128')]
    names = sys.argv[1:] or ['man']
    bufs = {n: bench.load_scene_bytes(n) for n in names}        # child processes before the GPU is touched
    out = {}
    for name in names:
        out[name] = {}
        for label, packet, budget in SETTINGS:
            os.environ['RTMI_PACKET'] = packet; os.environ['RTMI_DEFER_BUDGET'] = str(budget)
            os.environ['RTMI_SHARE_MIN_BUDGET'] = '0'
            ctx = capi.Context(0)
            sc = capi.DeviceScene(ctx, bufs[name]); bvh = capi.DeviceBVH(ctx, sc).build()
            cam = bench.demo_camera(bufs[name], 1920, 1080)
            dc = ctx.alloc(1920 * 1080 * 32)
            for _ in range(30):
                bvh.generate_contacts_primary_d(cam, dc)
            ctx.synchronize()
            ctx.set_profiling(True)
            for _ in range(20):
                bvh.generate_contacts_primary_d(cam, dc)
            k = ctx.trace_kernel_times()
            ctx.set_profiling(False)
            out[name][label] = dict(first_kernel_ms=round(k['lane_phase_ms'] / 20, 4), follow_up_ms=round(k['wave_phase_ms'] / 20, 4),
                                    counters=ctx.last_trace_counters())
            print(name, label, out[name][label], flush=True)
            ctx.free(dc); bvh.close(); sc.close(); ctx.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
