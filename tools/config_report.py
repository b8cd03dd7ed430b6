"""Timing of BASELINE.json configs 3, 4 and 5 (synthetic stand-ins) on one GPU: trace ms per frame, Mrays/s.
    python tools/config_report.py            # everything
    python tools/config_report.py config3    # only the Bunny-class BVH frame (the rocprofv3 pass of tools/profile_round.sh)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
import numpy as np
import gen_scenes
from openclraytracer_amd import capi
sys.path.insert(0, ROOT)
import bench
ONLY3 = sys.argv[1:] == ['config3']
# pack the scenes with the product's loader BEFORE the GPU is touched (no child processes afterwards)
for _name in ['synth_bunny'] + ([] if ONLY3 else ['synth_atrium']):
    if os.environ.get('RTMI_PACKED') == '1':      # under rocprofv3: the scenes come packed and NO child process is started
        if not os.path.exists('/tmp/_%s.bin' % _name):
            raise SystemExit('RTMI_PACKED=1 but /tmp/_%s.bin is missing: pack the scenes before the profiled pass (tools/profile_round.sh)' % _name)
        continue
    subprocess.check_call([os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool'), os.path.relpath(gen_scenes.ensure()[_name], ROOT),
                           '/tmp/_%s.bin' % _name], cwd=ROOT)
ctx = capi.Context(0)
out = {}
for name, res in [('synth_bunny', (1920, 1080))] + ([] if ONLY3 else [('synth_atrium', (1920, 1080)), ('synth_atrium', (3840, 2160))]):
    buf = open('/tmp/_%s.bin' % name, 'rb').read()
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, *res)
    npx = res[0] * res[1]
    dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    bvh.generate_contacts_primary_d(cam, dc); ctx.synchronize()      # calibrates
    tm = capi.Timer(ctx); K = 20 if ONLY3 else 5
    tm.start()
    for _ in range(K):
        bvh.generate_contacts_primary_d(cam, dc)
        capi.shade_blinn_phong_d(ctx, sc, cam, dc, npx, dr)
    tm.stop(); ms = tm.elapsed_ms() / K
    out['%s_%dx%d' % (name, res[0], res[1])] = dict(ms_per_frame=round(ms, 3), mrays_per_s=round(npx / ms / 1e3, 1),
                                                     triangles=sc.n_tris, depth=bvh.info()['max_depth'], strategy=bvh.trace_strategy())
    ctx.free(dc); ctx.free(dr); bvh.close(); sc.close()
# config 4: Two-Level Grid build + trace (+ shade) on the atrium, and on the bundled models for comparison with the BVH
for name, res in [] if ONLY3 else [('synth_atrium', (1920, 1080)), ('synth_bunny', (1920, 1080))] + [(s, (1920, 1080)) for s in bench.SCENES]:
    if name.startswith('synth'):
        buf = open('/tmp/_%s.bin' % name, 'rb').read()
    else:
        buf = bench.load_scene_bytes(name)
    sc = capi.DeviceScene(ctx, buf); g = capi.DeviceTLG(ctx, sc)
    tm = capi.Timer(ctx)
    g.build(); ctx.synchronize()
    tm.start()
    for _ in range(3):
        g.build()
    tm.stop(); build_ms = tm.elapsed_ms() / 3
    cam = bench.demo_camera(buf, *res)
    npx = res[0] * res[1]
    dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
    g.generate_contacts_primary_d(cam, dc); ctx.synchronize()
    K = 5
    tm.start()
    for _ in range(K):
        g.generate_contacts_primary_d(cam, dc)
    tm.stop(); trace_ms = tm.elapsed_ms() / K
    tm.start()
    for _ in range(K):
        g.generate_contacts_primary_d(cam, dc)
        capi.shade_blinn_phong_d(ctx, sc, cam, dc, npx, dr)
    tm.stop(); ms = tm.elapsed_ms() / K
    out['tlg_%s_%dx%d' % (name, res[0], res[1])] = dict(build_ms=round(build_ms, 3), trace_ms=round(trace_ms, 3), ms_per_frame=round(ms, 3),
                                                         mrays_per_s=round(npx / ms / 1e3, 1), triangles=sc.n_tris, info=g.info())
    ctx.free(dc); ctx.free(dr); g.close(); sc.close()
print(json.dumps(out, indent=1))
