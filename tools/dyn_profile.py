#!/usr/bin/env python3
"""The dynamic-scene frame (initializeFrame + construct + rt_render_frame, OpenCLRayTracerDemo.cpp:230-234 + 287-319) in a loop, host-timed; under
`rocprofv3 --kernel-trace` + tools/build_timeline.py: what such a frame consists of on the device.

    python tools/dyn_profile.py [scene] [--mode 0|2] [--k 20]        (RTMI_PACKED=1: the packed scene from /tmp, nothing spawned)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench  # noqa: E402
from build_profile import scene_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('scene', nargs='?', default='tenryuu')
    ap.add_argument('--mode', type=int, default=0)
    ap.add_argument('--k', type=int, default=20)
    ap.add_argument('--in-flight', type=int, default=1)
    a = ap.parse_args()
    buf = scene_bytes(a.scene)
    from openclraytracer_amd import capi
    ctx = capi.Context(0)
    if a.in_flight > 1:
        ctx.set_frames_in_flight(a.in_flight)
    sc = capi.DeviceScene(ctx, buf)
    bvh = capi.DeviceBVH(ctx, sc, mode=a.mode).build()
    cam = bench.demo_camera(buf, *bench.RES)
    npx = bench.RES[0] * bench.RES[1]
    dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    g = capi.FrameGraph(bvh, sc, cam, dc, dr)
    for _ in range(5):
        g.launch()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.k):
        g.launch()
    ctx.synchronize()
    static_ms = (time.perf_counter() - t0) * 1e3 / a.k
    g.close()

    def dyn():
        bvh.initialize_frame(); bvh.construct()
        capi.render_frame_d(bvh, sc, cam, dc, dr)
    for _ in range(4):
        dyn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.k):
        dyn()
    ctx.synchronize()
    dyn_ms = (time.perf_counter() - t0) * 1e3 / a.k
    t0 = time.perf_counter()
    for _ in range(a.k):
        bvh.build()
    ctx.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3 / a.k
    print('%s mode %d: static frame %.4f ms, build %.4f ms, build + frame %.4f ms (%.2f x the static frame)' % (a.scene, a.mode, static_ms, build_ms, dyn_ms, dyn_ms / static_ms))
    # one more, last in the trace
    dyn(); ctx.synchronize()


if __name__ == '__main__':
    main()
