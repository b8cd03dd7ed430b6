#!/usr/bin/env python3
"""construct() as a timed stage: rebuilds per scene and mode, host-timed (initializeFrame + construct, synchronous as the reference's), and
-- when run under `rocprofv3 --kernel-trace --stats` -- the per-kernel split of the same loop.

    python tools/build_profile.py [scene ...] [--mode 0|2] [--k 20] [--out file.json]
    scenes: the bundled names, synth_bunny, synth_atrium, heightfield (1 002 528 triangles, tests/test_gpu_configs.py)

Scenes are packed before the GPU is touched; with RTMI_PACKED=1 and the packed files in /tmp nothing is spawned (counter passes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
import bench  # noqa: E402


def scene_bytes(name):
    packed = '/tmp/_%s.bin' % name
    if os.environ.get('RTMI_PACKED') == '1':
        if not os.path.exists(packed):
            raise SystemExit('RTMI_PACKED=1 but %s is missing: pack the scenes first (tools/profile_round.sh)' % packed)
        return open(packed, 'rb').read()
    if name in bench.SCENES:
        buf = bench.load_scene_bytes(name)
    elif name == 'heightfield':
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from test_gpu_configs import _terrain_scene_bytes
        buf = _terrain_scene_bytes()
    else:
        import gen_scenes
        buf = bench.pack_scene_file(gen_scenes.ensure()[name])
    with open(packed, 'wb') as f:
        f.write(buf)
    return buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('scenes', nargs='*', default=bench.SCENES + ['synth_bunny', 'synth_atrium'])
    ap.add_argument('--mode', type=int, default=0)
    ap.add_argument('--k', type=int, default=20)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    bufs = {n: scene_bytes(n) for n in a.scenes}
    from openclraytracer_amd import capi
    ctx = capi.Context(0)
    out = {}
    for name, buf in bufs.items():
        sc = capi.DeviceScene(ctx, buf)
        bvh = capi.DeviceBVH(ctx, sc, mode=a.mode)
        for _ in range(3):
            bvh.build()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.k):
            bvh.build()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.k
        out[name] = dict(triangles=sc.n_tris, build_ms=round(ms, 4), mode=a.mode, reference_depth=bvh.info()['max_depth'], search_depth=bvh.search_tree()['depth'])
        print('%-14s %8d triangles  build %.4f ms' % (name, sc.n_tris, ms), file=sys.stderr, flush=True)
        bvh.close(); sc.close()
    ctx.close()
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
