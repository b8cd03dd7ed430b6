#!/usr/bin/env python3
"""Which frame sets the length of a step in which several scenes' frames are in flight?  Times the bench's timed region (one frame
graph per scene, launched back to back on the scenes' own streams, one synchronisation per step) for subsets of the bundled scenes.
   python tools/overlap_probe.py            (all five, each one left out in turn, each alone)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openclraytracer_amd import capi

RES = (1920, 1080)
names = ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
bufs = {n: bench.load_scene_bytes(n) for n in names}
order = os.environ.get('PROBE_ORDER', '0') == '1'
items = {}
for n in names:
    ctx = capi.Context(0)
    sc = capi.DeviceScene(ctx, bufs[n]); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(bufs[n], *RES)
    npx = RES[0] * RES[1]
    dc, do = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    ctx.set_frames_in_flight(1 if order else 5)
    g = capi.FrameGraph(bvh, sc, cam, dc, do)
    items[n] = dict(ctx=ctx, g=g, strategy=bvh.trace_strategy())
print({n: items[n]['strategy'] for n in names}, 'block order in graphs:', order, flush=True)


def run(sub, steps=300):
    for _ in range(20):
        for n in sub:
            items[n]['g'].launch()
    capi.device_synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for n in sub:
            items[n]['g'].launch()
    capi.device_synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


out = {'all': run(names)}
if os.environ.get('PROBE_QUICK') != '1':
    for n in names:
        out['without ' + n] = run([m for m in names if m != n])
    for n in names:
        out['only ' + n] = run([n])
    out['tenryuu + man'] = run(['tenryuu', 'man'])
    out['tenryuu + dragon'] = run(['tenryuu', 'clumsydragon'])
else:
    out['all again'] = run(names)
for k, v in out.items():
    print('%-24s %.4f ms per step' % (k, v), flush=True)
print(json.dumps(dict(what='frame graphs of the bundled scenes at 1920x1080, one per scene on its own stream, launched back to back; ms per step for subsets of the scenes', policy='latency' if order else 'throughput (rt_ctx_set_frames_in_flight(5))', strategies={n: items[n]['strategy'] for n in names}, ms_per_step={k: round(v, 4) for k, v in out.items()})))
