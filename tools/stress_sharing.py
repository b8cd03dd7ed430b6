"""Stress of the four-wide packet kernel's sub-tree sharing: many frames per scene with small budgets and small queues, the
contacts of every 50th frame compared with the first frame's (which the parity tests compare with the oracle).
    RTMI_PACKET=4 RTMI_DEFER_BUDGET=32 python tools/stress_sharing.py tenryuu man atrium"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from openclraytracer_amd import capi
frames = int(os.environ.get('STRESS_FRAMES', '1000'))
ctx = capi.Context(0)
for n in sys.argv[1:] or ['tenryuu', 'man']:
    buf = bench.load_atrium_bytes() if n == 'atrium' else bench.load_scene_bytes(n)
    res = (3840, 2160) if n == 'atrium' else (1920, 1080)
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, *res)
    npx = res[0] * res[1]
    dc = ctx.alloc(npx * 32)
    bvh.generate_contacts_primary_d(cam, dc)
    first = ctx.download(dc, npx, capi.CONTACT).tobytes()
    t0 = time.perf_counter()
    k = frames if n != 'atrium' else max(frames // 10, 20)
    for i in range(k):
        bvh.generate_contacts_primary_d(cam, dc)
        if i % 50 == 49:
            assert ctx.download(dc, npx, capi.CONTACT).tobytes() == first, (n, i)
    ctx.synchronize()
    print(n, k, 'frames ok, %.3f ms per frame' % ((time.perf_counter() - t0) / k * 1e3), bvh.trace_strategy(), ctx.last_trace_counters(), flush=True)
    ctx.free(dc); bvh.close(); sc.close()
