"""Diagnostics of the two traversal phases per bundled scene at 1920x1080: strategy, deferred rays, pool steps and nodes of the
wave-per-ray kernel (RTMI_TRACE_DIAG=1 for the last two; the counting perturbs the timings printed beside them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from openclraytracer_amd import capi
ctx = capi.Context(0)
for name in sys.argv[1:] or ['tenryuu', 'clumsydragon']:
    buf = bench.load_scene_bytes(name)
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, 1920, 1080)
    dc = ctx.alloc(1920 * 1080 * 32)
    bvh.generate_contacts_primary_d(cam, dc); ctx.synchronize()
    ctx.set_profiling(True)
    for _ in range(5):
        bvh.generate_contacts_primary_d(cam, dc)
    k = ctx.trace_kernel_times(); ctx.set_profiling(False)
    print(name, bvh.trace_strategy(), ctx.last_trace_counters(), 'first %.3f ms, wave-per-ray %.3f ms' % (k['lane_phase_ms'] / 5, k['wave_phase_ms'] / 5))
