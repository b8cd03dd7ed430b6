#!/usr/bin/env python3
"""When do the five streams of a short timed region finish?  Five scenes' frame graphs, K steps issued back to back, every stream's completion
time taken by a thread of its own -- for a device that has just idled (warm-up of W steps, as the driver's `--steps 20 --warmup 5`) and for one
that has been busy for a while.
   python tools/tail_probe.py [K] [W]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openclraytracer_amd import capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
RES = (1920, 1080)
names = os.environ.get('PROBE_NAMES', 'cube,gardenlight,tenryuu,man,clumsydragon').split(',')        # (the order in which the contexts -- streams -- are created and the frames issued)
bufs = {n: bench.load_scene_bytes(n) for n in names}
items = {}
FORCE = dict(kv.split(':') for kv in os.environ.get('PROBE_FORCE', '').split(',') if kv)      # e.g. clumsydragon:4/1048576 = RTMI_PACKET / RTMI_DEFER_BUDGET of that scene's context
KNOBS = set()
for n in names:
    for k in ('RTMI_PACKET', 'RTMI_DEFER_BUDGET') + tuple(KNOBS):
        os.environ.pop(k, None)
    if n in FORCE:
        os.environ['RTMI_PACKET'], os.environ['RTMI_DEFER_BUDGET'] = FORCE[n].split('/')
    for kv in os.environ.get('PROBE_ENV_' + n.upper(), '').split(','):            # e.g. PROBE_ENV_CLUMSYDRAGON=RTMI_LONG_BATCH=1 (knobs are read per context)
        if kv:
            os.environ[kv.split('=')[0]] = kv.split('=')[1]; KNOBS.add(kv.split('=')[0])
    ctx = capi.Context(0)
    # PROBE_PRIORITY=tenryuu:high,man:high,cube:low: that scene's frames on a stream of that priority (hipStreamCreateWithPriority)
    pr = dict(kv.split(':') for kv in os.environ.get('PROBE_PRIORITY', '').split(',') if kv).get(n)
    if pr:
        import ctypes as C
        hip = C.CDLL('libamdhip64.so')
        lo, hi = C.c_int(), C.c_int()
        assert hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)) == 0
        st = C.c_void_p()
        assert hip.hipStreamCreateWithPriority(C.byref(st), 1, {'high': hi.value, 'low': lo.value, 'normal': (lo.value + hi.value) // 2}[pr]) == 0      # 1 = non-blocking
        ctx.set_stream(st.value)
        print('stream priority', n, pr, (lo.value, hi.value), flush=True)
    ctx.set_frames_in_flight(5)
    sc = capi.DeviceScene(ctx, bufs[n]); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(bufs[n], *RES)
    npx = RES[0] * RES[1]
    dc, do = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    items[n] = dict(ctx=ctx, g=capi.FrameGraph(bvh, sc, cam, dc, do))
    if n in os.environ.get('PROBE_DOUBLE', '').split(','):          # a second stream for this scene: its frames alternate between two contexts (two frames in flight)
        ctx2 = capi.Context(0); ctx2.set_frames_in_flight(5)
        sc2 = capi.DeviceScene(ctx2, bufs[n]); bvh2 = capi.DeviceBVH(ctx2, sc2).build()
        items[n]['ctx2'] = ctx2
        items[n]['g2'] = capi.FrameGraph(bvh2, sc2, cam, ctx2.alloc(npx * 32), ctx2.alloc(npx * 16))


def region(steps, warm, label):
    for k in range(warm):
        for n in names:
            (items[n]['g2'] if (k & 1) and 'g2' in items[n] else items[n]['g']).launch()
    capi.device_synchronize()
    done = {}
    t0 = time.perf_counter()
    for k in range(steps):
        for n in names:
            (items[n]['g2'] if (k & 1) and 'g2' in items[n] else items[n]['g']).launch()
    issued = time.perf_counter() - t0

    def wait(n):
        items[n]['ctx'].synchronize()
        if 'ctx2' in items[n]:
            items[n]['ctx2'].synchronize()
        done[n] = time.perf_counter() - t0
    th = [threading.Thread(target=wait, args=(n,)) for n in names]
    for t in th: t.start()
    for t in th: t.join()
    end = max(done.values())
    print('%-28s %d steps: %.4f ms per step (issued in %.3f ms); streams done at %s ms' % (label, steps, end / steps * 1e3, issued * 1e3,
          {n: round(done[n] * 1e3, 2) for n in names}), flush=True)


time.sleep(0.3)
region(K, W, 'after 0.3 s of idling')
region(K, W, 'right behind the last')
region(300, 5, 'long')
region(K, W, 'right behind the long one')
time.sleep(0.3)
region(K, W, 'after 0.3 s of idling')
time.sleep(0.02)
region(K, W, 'after 20 ms of idling')
