import sys, os, numpy as np, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from openclraytracer_amd import capi
NAMES = ['tenryuu', 'man', 'atrium']
# every scene is packed BEFORE the GPU is touched (the packer is a child process: none is started behind a live HIP runtime)
bufs = {name: (bench.load_atrium_bytes() if name == 'atrium' else bench.load_scene_bytes(name)) for name in NAMES}
ctx = capi.Context(0)
out = {}
for name in NAMES:
    buf = bufs[name]
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, 1920, 1080)
    npx = 1920 * 1080
    d_r, d_c = ctx.alloc(npx * 48), ctx.alloc(npx * 32)
    capi.generate_primary_rays_d(ctx, cam, d_r)
    rays = ctx.download(d_r, npx, capi.RAY)
    y, x = np.divmod(np.arange(npx), 1920)
    # 8x8 blocks inside 64x64 tiles (the primary kernels' enumeration), pixels outside full tiles keep a coarse order
    key = ((y // 64) * 30 + (x // 64)) * 4096 + ((y % 64) // 8 * 8 + (x % 64) // 8) * 64 + (y % 8) * 8 + (x % 8)
    order = np.argsort(key, kind='stable')
    res = {}
    for tag, r in (('row_order', rays), ('block_order', rays[order].copy())):
        ctx.upload(d_r, r)
        for _ in range(3): bvh.generate_contacts_rays_d(d_r, npx, d_c)
        ctx.synchronize()
        tm = capi.Timer(ctx); tm.start()
        for _ in range(8): bvh.generate_contacts_rays_d(d_r, npx, d_c)
        tm.stop(); res[tag] = round(tm.elapsed_ms() / 8, 4); tm.close()
    out[name] = res
    ctx.free(d_r); ctx.free(d_c); bvh.close(); sc.close()
print(os.environ.get('RTMI_RAYS'), json.dumps(out))
