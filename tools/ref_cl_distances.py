#!/usr/bin/env python3
"""Measures, on the MI355X, how far the product (host-branch arithmetic) is from the reference's own ray kernels as ROCm
compiles them -- per bundled model at 320x240, for both builds of the reference's OpenCL C (cl_strict: IEEE divide / sqrt, no
contraction; cl_default: the reference's own options, BVHManager.cpp:86) and both kernels (generateContacts2 on a ray buffer,
generateContacts from the camera) -- and writes tests/golden/reference_cl_distances.json, the bounds tests/test_gpu_reference_cl.py
asserts.  Test infrastructure (uses oracle/): run on the GPU box, `python tools/ref_cl_distances.py`, copy the file back.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
RES = (320, 240)


def main():
    import oracle as orc
    import ref_cl
    from conftest import MODELS, packed_scene
    from openclraytracer_amd import capi
    out = {}
    ctx = capi.Context(0)
    for build in ('strict', 'default'):
        ref = ref_cl.RefCL(build)
        out[build] = {}
        for name in MODELS:
            buf = packed_scene(name)
            osc = orc.Scene(buf)
            ob = orc.BVH(osc, 0)
            cam = osc.demo_camera(*RES)
            rays = orc.generate_rays(cam)
            sc = capi.DeviceScene(ctx, buf)
            gb = capi.DeviceBVH(ctx, sc).build()
            row = {}
            for kernel, got, prod in (('generateContacts2', ref.generate_contacts2(rays, ob.nodes, ob.n, buf), gb.generate_contacts_rays(rays.view(capi.RAY))),
                                      ('generateContacts', ref.generate_contacts(cam, ob.nodes, ob.n, buf), gb.generate_contacts_primary(cam.view(capi.CAMERA)))):
                gh, ph = got['normal_t'][:, 3] > 0, prod['normalAndDistance'][:, 3] > 0
                both = gh & ph
                tp, tr = prod['normalAndDistance'][both, 3], got['normal_t'][both, 3]
                row[kernel] = dict(rays=int(len(got)), hits=int(gh.sum()), hit_mask_differs=int((gh != ph).sum()),
                                   material_differs=int((prod['materialIndex'][both] != got['materialIndex'][both]).sum()),
                                   t_bits_differ=int((tp.view(np.uint32) != tr.view(np.uint32)).sum()),
                                   t_rel_max=float((np.abs(tp - tr) / tr).max()) if both.any() else 0.0)
            out[build][name] = row
            print(build, name, row, flush=True)
            gb.close(); sc.close()
        ref.close()
    ctx.close()
    path = os.path.join(ROOT, 'gpurun_out', 'reference_cl_distances.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
    print('wrote', path)


if __name__ == '__main__':
    main()
