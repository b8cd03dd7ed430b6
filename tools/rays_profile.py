#!/usr/bin/env python3
"""The ray-buffer entry point under rocprofv3 (tools/profile_round.sh): bench.py's three legs per scene (the camera's rays as Ray[], the
reflection bounce, the shadow rays) with the library's own calibration, every leg verified against the oracle's digests; prints the
per-leg times.  Scenes come packed from /tmp (RTMI_PACKED=1): no child process."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    from openclraytracer_amd import capi
    fx = bench.bench_fixture()
    key = '%dx%d' % bench.RES
    scenes = [(n, bench.load_scene_bytes(n), fx['scenes'][n][key]['ray_buffer']) for n in bench.SCENES]
    buf = bench.load_atrium_bytes()
    if bench.sha256_hex(np.frombuffer(buf, dtype='u1')) == fx['atrium']['scene_sha256']:
        scenes.append(('atrium', buf, fx['atrium'][key]['ray_buffer']))
    ctx = capi.Context(0)
    out = {}
    for name, b, rec in scenes:
        sc = capi.DeviceScene(ctx, b)
        bvh = capi.DeviceBVH(ctx, sc).build()
        out[name] = bench.ray_buffer_leg(ctx, bvh, bench.demo_camera(b, *bench.RES), rec)
        bvh.close(); sc.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
