#!/usr/bin/env python3
"""What a frame costs beyond its kernels: the bench's step (captured frames of the bundled scenes, one stream each) at a resolution where the
device part is negligible, for 1, 2, 3 and 5 scenes in flight, next to the full-size step.  Host time to issue a step, and the step itself."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
bufs = {n: bench.load_scene_bytes(n) for n in bench.SCENES}          # packed before the GPU is touched
from openclraytracer_amd import capi
out = {}
for res in ((64, 48), (1920, 1080)):
    work = []
    for n in bench.SCENES:
        ctx = capi.Context(0)
        ctx.set_frames_in_flight(len(bench.SCENES))
        sc = capi.DeviceScene(ctx, bufs[n]); bvh = capi.DeviceBVH(ctx, sc).build()
        cam = bench.demo_camera(bufs[n], *res)
        npx = res[0] * res[1]
        dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
        bvh.generate_contacts_primary_d(cam, dc); ctx.synchronize()
        work.append(dict(name=n, ctx=ctx, g=capi.FrameGraph(bvh, sc, cam, dc, dr), strategy=bvh.trace_strategy()))
    for sel in ([0], [1], [2], [3], [4], [0, 1], [0, 1, 2], [0, 1, 2, 3, 4]):
        ws = [work[i] for i in sel]
        for _ in range(50):
            for w in ws: w['g'].launch()
        for w in ws: w['ctx'].synchronize()
        K = 300
        t0 = time.perf_counter()
        for _ in range(K):
            for w in ws: w['g'].launch()
        t1 = time.perf_counter()
        for w in ws: w['ctx'].synchronize()
        t2 = time.perf_counter()
        out['%dx%d %s' % (res[0], res[1], '+'.join(work[i]['name'] for i in sel))] = dict(
            host_issue_ms_per_step=round((t1 - t0) / K * 1e3, 4), ms_per_step=round((t2 - t0) / K * 1e3, 4), last_launch=[w['g'].last_launch() for w in ws][:1])
    out['strategies %dx%d' % res] = {w['name']: w['strategy'] for w in work}
print(json.dumps(out, indent=1))
