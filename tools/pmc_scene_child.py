#!/usr/bin/env python3
"""Profiled child of tools/p4_wait_pmc.sh: one packed scene, one calibration frame, then N isolated traversal launches whose kernels
carry the template tag <1, 2> (rt_ctx_set_profiling(ctx, 2)).  Starts no process of its own (it runs under rocprofv3 --pmc).
    python tools/pmc_scene_child.py packed.bin resX resY launches
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/tools')


def main():
    import numpy as np
    from openclraytracer_amd import capi
    import bench
    buf = open(sys.argv[1], 'rb').read()
    resx, resy, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ctx = capi.Context(0)
    scene = capi.DeviceScene(ctx, buf)
    bvh = capi.DeviceBVH(ctx, scene).build()
    cam = bench.demo_camera(buf, resx, resy)
    dc = ctx.alloc(resx * resy * 32)
    bvh.generate_contacts_primary_d(cam, dc)
    ctx.synchronize()
    for _ in range(20):                       # clocks
        bvh.generate_contacts_primary_d(cam, dc)
    ctx.synchronize()
    ctx.set_profiling(2)
    for _ in range(n):
        bvh.generate_contacts_primary_d(cam, dc)
    ctx.synchronize()
    ctx.set_profiling(False)
    print('CHILD', bvh.trace_strategy(), flush=True)
    ctx.free(dc); bvh.close(); scene.close(); ctx.close()


if __name__ == '__main__':
    main()
