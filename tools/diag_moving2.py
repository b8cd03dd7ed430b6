#!/usr/bin/env python3
"""development diagnostic: the bench's sequence for one scene, per-pose replay times"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import bench
from openclraytracer_amd import capi


def log(*a):
    print(*a, flush=True)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'clumsydragon'
    do_rebuild = '--rebuild' in sys.argv
    do_prof = '--prof' in sys.argv
    buf = bench.load_scene_bytes(name)
    ctx = capi.Context(0)
    sc = capi.DeviceScene(ctx, buf)
    bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, 1920, 1080)
    npx = 1920 * 1080
    dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    bvh.generate_contacts_primary_d(cam, dc); ctx.synchronize()
    g = capi.FrameGraph(bvh, sc, cam, dc, dr)
    log(name, bvh.trace_strategy())
    for _ in range(20):
        g.launch()
    ctx.synchronize()
    if do_prof:
        ctx.set_profiling(2)
        for _ in range(10):
            bvh.generate_contacts_primary_d(cam, dc)
        log('kernel times', ctx.trace_kernel_times())
        ctx.set_profiling(False)
    if do_rebuild:
        for _ in range(10):
            bvh.initialize_frame(); bvh.construct()
            bvh.generate_contacts_primary_d(cam, dc)
            capi.shade_blinn_phong_frame_d(ctx, sc, cam, dc, dr)
        ctx.synchronize()
        log('after rebuilds', bvh.trace_strategy())
        g.close()
        g = capi.FrameGraph(bvh, sc, cam, dc, dr)
    poses = bench.moving_cameras(buf, 1920, 1080, 32)
    for k in range(32):
        g.set_camera(poses[k:k + 1]); g.launch()
    ctx.synchronize()
    for k in range(32):
        g.set_camera(poses[k:k + 1]); g.launch(); ctx.synchronize()
        k1 = g.last_launch()
        t = time.perf_counter()
        for _ in range(4):
            g.launch()
        ctx.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / 4
        log('pose %2d  first kind %d  then kind %d  %.4f ms per replay' % (k, k1, g.last_launch(), ms), ctx.last_trace_counters())
    log('done')


if __name__ == '__main__':
    main()
