#!/usr/bin/env python3
"""Generates tests/golden/bench_frames.json: what bench.py verifies its timed frames against, in the run that times them.

    make -C oracle && python tests/golden/make_golden_bench.py          (repo root; CPU only, a few minutes on 8 cores)

All values come from the CPU oracle (oracle/, pinned on the reference's own kernels: tests/golden/reference_cl.json).  Per bundled
scene at 1920x1080, and for the Sponza-class stand-in (assets/synth) at 1920x1080 and 3840x2160:
  default / pose<k>   camera bits, SHA-256 of the 32-byte contacts, hit count, per-channel sums of the shaded RGBA (float64) and every
                      SAMPLE_STRIDE-th pixel's RGBA bits.  pose<k> are poses of bench.py's moving-camera leg (bench.moving_cameras);
                      the fixture carries their camera bits, so the bench renders exactly these cameras whatever its host's libm does.
  ray_buffer          the ray-buffer entry point (generateContacts(CLBuffer& rays, ..), BVHManager.cpp:292-329): the camera's own rays
                      handed over as Ray[], the reflection bounce and the shadow rays towards the scene's light generated from the
                      primary contacts (rt_generate_secondary_rays' specification, oracle/rt_oracle.c): ray count, SHA-256 of the rays
                      and of the contacts, hits.
The stand-in scene is generated procedurally; the fixture holds the SHA-256 of its packed scene buffer so that a host that
generates other bytes is told apart from a wrong frame.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
os.chdir(ROOT)
import oracle as orc  # noqa: E402
import bench  # noqa: E402   (camera poses of the moving-camera leg; nothing of the GPU path is touched)

OUT = os.path.join(ROOT, 'tests', 'golden', 'bench_frames.json')
SAMPLE_STRIDE = 4999
POSES = (5, 17)                 # of bench.moving_cameras(.., 32); pose 0 is the default camera
LIGHT = (0.0, 10.0, -330.0)     # assets/scenes/*.txt: LIGHT 0.0 10.0 -330.0 1000.0
EPS = 1e-3


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def key_bytes(c):
    """what RT_MODE_DEVICE_ARITH pins bit for bit against the reference's own kernel: pixelIndex, materialIndex, t (the normal comes out of
    another normalize there)"""
    t = np.ascontiguousarray(c['normal_t'][:, 3]).view(np.uint32)
    return np.stack([c['pixelIndex'], c['materialIndex'], t], axis=1).astype('<u4').tobytes()


def frame_record(osc, ob, cam, modes=False):
    c, st = ob.trace_primary(cam)
    rgba = orc.shade_frame(osc, cam, c)
    extra = {}
    if modes:
        # RT_MODE_FAST keeps hit mask and t (another tie order: normal / material may differ where two triangles tie on t)
        extra['t_sha256'] = sha(np.ascontiguousarray(c['normal_t'][:, 3]).tobytes())
        # RT_MODE_DEVICE_ARITH: dot / cross fused as ROCm's OpenCL C library fuses them (the oracle's switch, pinned on the reference kernels)
        orc.set_device_arith(True)
        try:
            cd, std = ob.trace_primary(cam)
        finally:
            orc.set_device_arith(False)
        extra['device_arith'] = dict(key_sha256=sha(key_bytes(cd)), hits=int(std.hits))
    return c, dict(**extra, camera_bits=[int(x) for x in cam.view(np.uint32).ravel()], rays=len(c), hits=int(st.hits),
                   contacts_sha256=sha(c.tobytes()), rgba_sum=[float(x) for x in rgba.astype(np.float64).sum(axis=0)],
                   rgba_samples_bits=[[int(v) for v in px] for px in rgba[::SAMPLE_STRIDE].view(np.uint32)])


def ray_buffer_record(ob, cam, primary_contacts):
    out = {}
    rays0 = orc.generate_rays(cam)
    legs = (('primary', rays0),
            ('reflect', orc.generate_secondary_rays(primary_contacts, 1, eps=EPS, rays_in=rays0)),
            ('shadow', orc.generate_secondary_rays(primary_contacts, 0, LIGHT, EPS, rays_in=rays0)))
    for leg, rays in legs:
        c, st = ob.trace_rays(rays)
        out[leg] = dict(rays=len(rays), rays_sha256=sha(rays.tobytes()), contacts_sha256=sha(c.tobytes()), hits=int(st.hits),
                        inner_visits=int(st.inner_visits), leaf_tests=int(st.leaf_tests))
    return out


def main():
    t0 = time.time()
    fx = dict(sample_stride=SAMPLE_STRIDE, light=list(LIGHT), eps=EPS, scenes={})
    res = bench.RES
    key = '%dx%d' % res
    for name in bench.SCENES:
        buf = bench.load_scene_bytes(name)
        osc = orc.Scene(buf)
        ob = orc.BVH(osc, 0)
        cam = osc.demo_camera(*res)
        assert cam.tobytes() == bench.demo_camera(buf, *res).tobytes(), 'bench.demo_camera differs from the oracle camera'
        c, rec = frame_record(osc, ob, cam, modes=True)
        entry = dict(scene_sha256=sha(buf), default=rec, ray_buffer=ray_buffer_record(ob, cam, c))
        poses = bench.moving_cameras(buf, res[0], res[1], 32)
        for k in POSES:
            _, entry['pose%d' % k] = frame_record(osc, ob, poses[k:k + 1].view(orc.CAMERA))
        fx['scenes'][name] = {key: entry}
        print('%-13s done  %.0f s' % (name, time.time() - t0), flush=True)
    buf = bench.load_atrium_bytes()
    osc = orc.Scene(buf)
    ob = orc.BVH(osc, 0)
    atrium = dict(scene_sha256=sha(buf), triangles=osc.n_tris)
    cam = osc.demo_camera(*res)
    c, rec = frame_record(osc, ob, cam)
    atrium[key] = dict(default=rec, ray_buffer=ray_buffer_record(ob, cam, c))
    print('atrium %s done  %.0f s' % (key, time.time() - t0), flush=True)
    cam5 = osc.demo_camera(*bench.RES5)
    assert cam5.tobytes() == bench.demo_camera(buf, *bench.RES5).tobytes()
    _, rec5 = frame_record(osc, ob, cam5)
    atrium['%dx%d' % bench.RES5] = dict(default=rec5)
    # BASELINE.json configs[3]: the Two-Level Grid (densities 2.0 / 2.0) over the same stand-in, 1920x1080
    tg = orc.TLG(osc)
    ct, stt = tg.trace_primary(cam)
    rgba_t = orc.shade_frame(osc, cam, ct)
    atrium[key]['tlg'] = dict(contacts_sha256=sha(ct.tobytes()), hits=int(stt['hits']), info={k: int(v) for k, v in tg.info.items()},
                              rgba_sum=[float(x) for x in rgba_t.astype(np.float64).sum(axis=0)],
                              rgba_samples_bits=[[int(v) for v in px] for px in rgba_t[::SAMPLE_STRIDE].view(np.uint32)])
    print('atrium grid done  %.0f s' % (time.time() - t0), flush=True)
    fx['atrium'] = atrium
    # BASELINE.json configs[2]: the Bunny-class stand-in (69 564 triangles), BVH, 1920x1080
    buf = bench.load_bunny_bytes()
    osc = orc.Scene(buf)
    ob = orc.BVH(osc, 0)
    cam = osc.demo_camera(*res)
    assert cam.tobytes() == bench.demo_camera(buf, *res).tobytes()
    _, recb = frame_record(osc, ob, cam)
    fx['bunny'] = {'scene_sha256': sha(buf), 'triangles': osc.n_tris, key: dict(default=recb)}
    print('bunny done  %.0f s' % (time.time() - t0), flush=True)
    json.dump(fx, open(OUT, 'w'), indent=None, separators=(',', ':'), sort_keys=True)
    print('wrote %s (%.1f KB) in %.0f s' % (OUT, os.path.getsize(OUT) / 1024.0, time.time() - t0))


if __name__ == '__main__':
    main()
