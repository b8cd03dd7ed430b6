#!/usr/bin/env python3
"""Rebuild soak: thousands of construct() calls per scene and mode on one allocation (the replayed build graph), the node bytes of both
hierarchies, the depths and a traced frame compared with the first build's at intervals -- a race in the chain (span refit, the last
kernel's ticket, the depth words written to pinned memory) would show as a difference.

    python tools/build_soak.py [--builds 3000] [--every 50]
"""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench  # noqa: E402
from build_profile import scene_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--builds', type=int, default=3000)
    ap.add_argument('--every', type=int, default=50)
    ap.add_argument('--stress', action='store_true', help='frames of another scene in flight on another stream while the builds run')
    ap.add_argument('scenes', nargs='*', default=bench.SCENES + ['synth_atrium'])
    a = ap.parse_args()
    bufs = {n: scene_bytes(n) for n in a.scenes}
    # (the noise scene as well: every packer child runs before the first HIP call of this process)
    nbuf = None
    if a.stress:
        nbuf = bench.load_scene_bytes('man') if os.environ.get('RTMI_PACKED') != '1' else scene_bytes('man')
    from openclraytracer_amd import capi
    ctx = capi.Context(0)
    noise = None
    if a.stress:
        nctx = capi.Context(0)
        nsc = capi.DeviceScene(nctx, nbuf); nbvh = capi.DeviceBVH(nctx, nsc).build()
        ncam = bench.demo_camera(nbuf, *bench.RES)
        npx = bench.RES[0] * bench.RES[1]
        noise = capi.FrameGraph(nbvh, nsc, ncam, nctx.alloc(npx * 32), nctx.alloc(npx * 16))
    bad = 0
    for name, buf in bufs.items():
        sc = capi.DeviceScene(ctx, buf)
        cam = bench.demo_camera(buf, 640, 360)
        for mode in (0, 2):
            bvh = capi.DeviceBVH(ctx, sc, mode=mode).build()

            def state():
                h = hashlib.sha256()
                h.update(bvh.read_nodes().tobytes())
                h.update(bvh.read_search_nodes().tobytes())
                h.update(repr((bvh.info()['max_depth'], bvh.search_tree()['depth'])).encode())
                h.update(bvh.generate_contacts_primary(cam).tobytes())
                return h.hexdigest()
            first = state()
            builds = a.builds if sc.n_tris < 100000 else max(a.builds // 6, 1)
            diffs = 0
            depths = (bvh.info()['max_depth'], bvh.search_tree()['depth'])
            for k in range(1, builds + 1):
                if noise is not None:
                    for _ in range(3):
                        noise.launch()                      # (another stream's frames in flight while this one builds)
                bvh.build()
                if (bvh.info()['max_depth'], bvh.search_tree()['depth']) != depths:      # the words the build's last kernel writes to pinned memory, every build
                    diffs += 1
                elif k % a.every == 0 and state() != first:
                    diffs += 1
            print('%-14s mode %d: %d rebuilds, compared every %d: %s' % (name, mode, builds, a.every, 'all equal' if diffs == 0 else '%d DIFFER' % diffs), flush=True)
            bad += diffs
            bvh.close()
        sc.close()
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
