#!/usr/bin/env python3
"""Steps per 8x8 block of the four-wide packet kernel (its own count, rt_bvh_block_order) for a bundled scene at 1920x1080:
writes gpurun_out/block_cost_<scene>.npy (uint32 per block in work-item order: 64x64 tiles row-major, 8x8 blocks row-major inside)
   python tools/block_costs.py man tenryuu"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openclraytracer_amd import capi

os.environ['RTMI_PACKET'] = '4'; os.environ['RTMI_DEFER_BUDGET'] = str(1 << 20)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
names = sys.argv[1:] or ['man']
bufs = {n: bench.load_scene_bytes(n) for n in names}
for name in names:
    ctx = capi.Context(0)
    sc = capi.DeviceScene(ctx, bufs[name]); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(bufs[name], 1920, 1080)
    dc = ctx.alloc(1920 * 1080 * 32)
    for _ in range(3):
        bvh.generate_contacts_primary_d(cam, dc)
    ctx.synchronize()
    order, cost = bvh.block_order()
    steps = cost & 0xFFFF
    np.save(os.path.join(ROOT, 'gpurun_out', 'block_cost_%s.npy' % name), cost)
    np.save(os.path.join(ROOT, 'gpurun_out', 'search_nodes_%s.npy' % name), bvh.read_search_nodes())
    print(name, 'blocks', len(cost), 'steps total', int(steps.sum()), 'mean %.1f median %d p90 %d p99 %d max %d' % (
        steps.mean(), np.median(steps), np.percentile(steps, 90), np.percentile(steps, 99), steps.max()), 'search tree', bvh.search_tree(), flush=True)
    ctx.free(dc); bvh.close(); sc.close(); ctx.close()
