#!/usr/bin/env python3
"""development diagnostic: replay times of a frame graph before / after rt_frame_graph_set_camera, per strategy"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import bench
from openclraytracer_amd import capi


def log(*a):
    print(*a, flush=True)


def timed(ctx, fn, n):
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t) * 1e3 / n


def main():
    names = sys.argv[1:] or ['clumsydragon']
    items = []
    for name in names:
        buf = bench.load_scene_bytes(name)
        items.append((name, buf))
    work = []
    for name, buf in items:
        ctx = capi.Context(0)
        sc = capi.DeviceScene(ctx, buf)
        bvh = capi.DeviceBVH(ctx, sc).build()
        cam = bench.demo_camera(buf, 1920, 1080)
        npx = 1920 * 1080
        dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
        g = capi.FrameGraph(bvh, sc, cam, dc, dr)
        poses = bench.moving_cameras(buf, 1920, 1080, 32)
        log(name, 'strategy', bvh.trace_strategy(), 'calibration launches', bvh.calibration_launches(), 'gpu ms %.3f' % bvh.calibration_gpu_ms())
        work.append(dict(name=name, ctx=ctx, sc=sc, bvh=bvh, cam=cam, g=g, poses=poses, dc=dc, dr=dr))
    for it in work:
        ctx, g, poses = it['ctx'], it['g'], it['poses']
        log(it['name'], 'static replays        %.4f ms' % timed(ctx, g.launch, 20), 'kind', g.last_launch())
        g.set_camera(poses[1:2]); g.launch(); ctx.synchronize()
        log(it['name'], 'after set_camera: first launch kind', g.last_launch())
        log(it['name'], 'replays after update  %.4f ms' % timed(ctx, g.launch, 20), 'kind', g.last_launch())
        log(it['name'], 'replays again         %.4f ms' % timed(ctx, g.launch, 20), 'kind', g.last_launch())
        g.set_camera(poses[2:3]); g.launch(); ctx.synchronize()
        for k in range(5):
            log(it['name'], '  single replay %d     %.4f ms' % (k, timed(ctx, g.launch, 1)), 'kind', g.last_launch())
        def mv(state=[0]):
            state[0] += 1
            g.set_camera(poses[state[0] % 32:state[0] % 32 + 1]); g.launch()
        log(it['name'], 'moving, no sync       %.4f ms' % timed(ctx, mv, 64), 'kind', g.last_launch())
        def mvs(state=[0]):
            state[0] += 1
            g.set_camera(poses[state[0] % 32:state[0] % 32 + 1]); g.launch(); ctx.synchronize()
        log(it['name'], 'moving, sync each     %.4f ms' % timed(ctx, mvs, 32), 'kind', g.last_launch())
        def eager(state=[0]):
            state[0] += 1
            p = poses[state[0] % 32:state[0] % 32 + 1]
            it['bvh'].generate_contacts_primary_d(p, it['dc']); capi.shade_blinn_phong_frame_d(ctx, it['sc'], p, it['dc'], it['dr'])
        log(it['name'], 'moving eager          %.4f ms' % timed(ctx, eager, 32))
    if len(work) > 1:
        log('all streams, moving, 64 steps ...')
        for it in work:
            it['ctx'].synchronize()
        t = time.perf_counter()
        for k in range(64):
            for it in work:
                it['g'].set_camera(it['poses'][k % 32:k % 32 + 1]); it['g'].launch()
            if k % 8 == 7:
                log('  issued step', k)
        for it in work:
            it['ctx'].synchronize(); log('  synced', it['name'])
        log('all streams moving: %.4f ms per step' % ((time.perf_counter() - t) * 1e3 / 64))
    for it in work:
        it['g'].close()
    log('done')


if __name__ == '__main__':
    main()
