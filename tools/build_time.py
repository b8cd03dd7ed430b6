import os, sys, time, json
sys.path.insert(0, '/root/repo')
import bench
from openclraytracer_amd import capi
ctx = capi.Context(0)
out = {}
for n in ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']:
    buf = bench.load_scene_bytes(n)
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    for _ in range(3): bvh.build()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(20): bvh.build()
    ctx.synchronize()
    out[n] = round((time.perf_counter() - t) / 20 * 1e3, 3)
    bvh.close(); sc.close()
print(os.environ.get('RTMI_REFIT', 'default'), json.dumps(out))
