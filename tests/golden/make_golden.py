#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run from the repo root, in the authoring
container where /root/reference and oracle/_ref exist):

    make -C oracle && make -C oracle ref && python tests/golden/make_golden.py

What pins what
  tinyobj.json          sha256 of the REFERENCE's own OBJ parser output (oracle/_ref/tinyobj_dump, i.e. the
                        vendored tiny_obj_loader.h compiled from /root/reference) for every bundled model:
                        real reference-derived known answers for the loader stage.
  reference_probe.json  numbers SURVEY.md records from the reference's arithmetic headers compiled during the
                        survey (hit counts, visits/ray, leaf tests/ray, max stack at 640x480): the only
                        reference-derived known answers for the traversal stage (written by hand, see file).
  scenes.json, bvh.json, frames.json, contacts_64x48.npz, kat.json
                        outputs of the repo's CPU oracle (oracle/): regression pins and the counter fixture
                        SURVEY 8(d) asks for (inner visits I, leaf tests L per ray under reference semantics).
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
os.chdir(ROOT)
import oracle as orc  # noqa: E402
import scene_pack as sp  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
MODELS = {'cube': 'assets/testmodels/Cube/cube.obj',
          'gardenlight': 'assets/testmodels/Gardenlight/gardenlight01.obj',
          'tenryuu': 'assets/testmodels/Tenryuu/light-cruiser-tenryuu.obj',
          'man': 'assets/testmodels/ManFigure/standard-male-figure.obj',
          'clumsydragon': 'assets/testmodels/ClumsyDragon/clumsy-dragon-vray.obj'}


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    tiny, scenes, bvhs, frames, small = {}, {}, {}, {}, {}
    dumper = os.path.join(ROOT, 'oracle', '_ref', 'tinyobj_dump')
    for name, obj in MODELS.items():
        if os.path.exists(dumper):
            tmp = '/tmp/_tobj_%s.bin' % name
            subprocess.check_call([dumper, obj, tmp])
            data = open(tmp, 'rb').read()
            shapes, mats = sp.load_tinyobj_dump(tmp)
            tiny[name] = dict(sha256=sha(data), bytes=len(data), shapes=len(shapes), materials=len(mats),
                              triangles=int(sum(len(s.indices) // 3 for s in shapes)),
                              max_vertices=int(max(len(s.positions) // 3 for s in shapes)))
        buf = sp.load_scene('assets/scenes/%s.txt' % name)
        hdr = sp.scene_header(buf)
        scenes[name] = dict(sha256=sha(buf), bytes=len(buf),
                            **{k: int(v) for k, v in hdr.items() if not k.startswith('bbox')},
                            bbox_min_bits=[int(np.float32(x).view(np.uint32)) for x in hdr['bbox_min']],
                            bbox_max_bits=[int(np.float32(x).view(np.uint32)) for x in hdr['bbox_max']])
        osc = orc.Scene(buf)
        b = orc.BVH(osc, 0)
        rc, depth = b.check_hierarchy()
        assert rc == 0
        bvhs[name] = dict(leaves=b.n, morton_items=b.N, depth=depth, pairs_sha256=sha(b.pairs.tobytes()),
                          nodes_sha256=sha(b.nodes.tobytes()))
        frames[name] = {}
        for res in [(64, 48), (640, 480), (1920, 1080)]:
            cam = osc.demo_camera(*res)
            c, st = b.trace_primary(cam)
            rgba = orc.shade_frame(osc, cam, c)
            n = len(c)
            frames[name]['%dx%d' % res] = dict(
                camera_bits=[int(x) for x in cam.view(np.uint32).ravel()],
                rays=n, hits=int(st.hits), inner_visits=int(st.inner_visits), leaf_tests=int(st.leaf_tests),
                max_stack=int(st.max_stack), sum_t=float(st.sum_t), contacts_sha256=sha(c.tobytes()),
                rgba_sum=[float(x) for x in rgba.astype(np.float64).sum(axis=0)])
            if res == (64, 48):
                small[name + '_contacts'] = c.view(np.uint8).reshape(-1, 32)
                small[name + '_rgba'] = rgba
    # scan / sort known-answer tests: inputs are regenerated from the seed in the tests
    kat = {}
    rng = np.random.default_rng(1234)
    for n in [2, 64, 1024, 65536]:
        a = rng.integers(0, 1 << 10, size=n, dtype=np.uint32)
        kat['scan_%d' % n] = sha(orc.scan_inclusive(a).tobytes())
        pairs = np.zeros(n, dtype=orc.PAIR)
        pairs['key'] = rng.integers(0, 16, size=n, dtype=np.uint32)          # 16 distinct keys: stresses tie order
        pairs['value'] = np.arange(n, dtype=np.uint32)
        kat['sort_%d' % n] = sha(orc.bitonic_sort(pairs).tobytes())
    if tiny:
        json.dump(tiny, open(os.path.join(OUT, 'tinyobj.json'), 'w'), indent=1, sort_keys=True)
    json.dump(scenes, open(os.path.join(OUT, 'scenes.json'), 'w'), indent=1, sort_keys=True)
    json.dump(bvhs, open(os.path.join(OUT, 'bvh.json'), 'w'), indent=1, sort_keys=True)
    json.dump(frames, open(os.path.join(OUT, 'frames.json'), 'w'), indent=1, sort_keys=True)
    json.dump(kat, open(os.path.join(OUT, 'kat.json'), 'w'), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'contacts_64x48.npz'), **small)
    print('wrote fixtures to', OUT)


if __name__ == '__main__':
    main()
