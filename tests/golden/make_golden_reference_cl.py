"""Generates tests/golden/reference_cl.json ON THE GPU BOX: outputs of the REFERENCE's own OpenCL C kernels (compiled from
/root/reference into oracle/_ref/cl_strict/*.co by `make -C oracle ref_cl`, run through oracle/ref_cl.py on the MI355X).

    gpurun -- 'python tests/golden/make_golden_reference_cl.py gpurun_out/reference_cl.json'
    cp gpurun_out/reference_cl.json tests/golden/reference_cl.json

The fixture holds SHA-256 digests (and small verbatim samples) of what the reference kernels produced, together with
digests of the inputs they were given, so that tests/test_cpu_reference_pins.py can check the CPU oracle against the
reference without a GPU and without /root/reference.  Every input is produced by committed code from committed assets
(assets/scenes/*.txt through oracle/scene_pack.py; rays through orc.generate_rays; node arrays through the oracle's
build), so the digests can be recomputed anywhere.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
import oracle as orc          # noqa: E402
import ref_cl                 # noqa: E402
import scene_pack             # noqa: E402

MODELS = ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
RES = (320, 240)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def contact_key(c) -> np.ndarray:
    """what is compared bit for bit: pixelIndex, materialIndex, t (the normal goes through fast_normalize, an approximation)"""
    k = np.zeros((len(c), 3), dtype=np.uint32)
    k[:, 0] = c['pixelIndex']
    k[:, 1] = c['materialIndex']
    k[:, 2] = c['normal_t'][:, 3].view(np.uint32)
    return k


def tlg_rays(cam):
    rays = orc.generate_rays(cam)
    d = rays['direction'][:, :3]
    rays = rays[(np.isfinite(d) & (d != 0)).all(axis=1)].copy()     # the reference DDA does not terminate on the others
    rays['idx'] = np.arange(len(rays))
    return rays


def main(out_path):
    os.chdir(ROOT)
    ref = ref_cl.RefCL('strict')
    out = {'_about': 'outputs of the reference OpenCL C kernels on gfx950 (oracle/_ref/cl_strict), see make_golden_reference_cl.py',
           'resolution': list(RES), 'models': {}}
    rng = np.random.default_rng(1234)
    out['scan'] = {}
    for size in (2, 64, 512):
        v = rng.integers(0, 1024, size=size, dtype=np.uint32)
        out['scan'][str(size)] = {'input': v.tolist(), 'output': ref.prefix_sum_one_group(v).tolist()}
    for name in MODELS:
        buf = scene_pack.load_scene(os.path.join('assets', 'scenes', name + '.txt'))
        sc = orc.Scene(buf)
        F, Lz = orc.build_stages(sc, 0), orc.build_stages(sc, 1)
        n, N = F['n'], F['N']
        m = {'n': n, 'N': N, 'scene_sha': hashlib.sha256(buf).hexdigest()}
        nodes, pairs = ref.calculate_morton_codes(buf, n)
        m['morton_nodes_sha'], m['morton_pairs_sha'] = sha(nodes), sha(pairs)
        srt = ref.bitonic_sort(pairs)
        m['sorted_pairs_sha'], m['sort_launches'] = sha(srt), ref.sort_launches
        rad = ref.build_radix_tree(nodes, srt, n)            # the reference's own chain: its leaves, its sorted pairs
        m['radix_literal_nodes_sha'] = sha(rad)
        if n <= 400:
            m['sorted_pairs'] = srt.view(np.uint32).reshape(-1, 2).tolist()
            m['radix_literal_nodes_hex'] = rad.tobytes().hex()
        # refit: the reference kernel on the complete (FIXED) tree -- racy on this hardware, recorded as such
        box = ref.compute_bounding_boxes(F['radix'], n)
        wrong = (box.view(np.uint8).reshape(len(box), -1) != F['nodes'].view(np.uint8).reshape(len(box), -1)).any(axis=1)
        inside = bool(((box['lo'][:, :3] >= F['nodes']['lo'][:, :3]) & (box['hi'][:, :3] <= F['nodes']['hi'][:, :3])).all())
        m['refit'] = {'nodes': int(len(box)), 'differing_from_fixed': int(wrong.sum()), 'all_boxes_inside_fixed_boxes': inside}
        cam = sc.demo_camera(*RES)
        rays = orc.generate_rays(cam)
        m['rays_sha'] = sha(rays)
        m['trace'] = {}
        for mode, st in ((0, F), (1, Lz)):
            ob = orc.BVH(sc, mode)
            _, stats = ob.trace_primary(cam)
            assert stats.max_stack < 30, 'the reference stack holds 32 entries'
            got = ref.generate_contacts2(rays, st['nodes'], n, buf)
            hit = got['normal_t'][:, 3] > 0
            m['trace'][str(mode)] = {'nodes_sha': sha(st['nodes']), 'key_sha': sha(contact_key(got)), 'hits': int(hit.sum()),
                                     'sum_t': float(got['normal_t'][hit, 3].astype(np.float64).sum()),
                                     'sample': contact_key(got)[::997].tolist(),
                                     'sample_normals': got['normal_t'][::997, :3].astype(np.float64).round(9).tolist()}
        prim = ref.generate_contacts(cam, F['nodes'], n, buf)     # generateRay normalises with fast_normalize: not bit-defined
        ph = prim['normal_t'][:, 3] > 0
        m['primary'] = {'hits': int(ph.sum()), 'mask_sha': sha(ph), 'material_sha': sha(prim['materialIndex'])}
        # Two-Level Grid: the kernels that stay in range on a consistent grid
        t = orc.TLG(sc)
        g = {'grid_sha': sha(t.grid), 'top_cells_sha': sha(t.top_cells), 'leaf_ranges_sha': sha(t.leaf_ranges), 'leaf_pairs_sha': sha(t.leaf_pairs)}
        g['prepare_counts_sha'] = sha(ref.tlg_prepare_data(buf, t.grid, n))
        top, leaves = ref.tlg_fill_top_level_cells(t.cell_ranges, t.grid)
        g['cell_ranges_sha'], g['cell_res_sha'], g['cell_leaf_counts_sha'] = sha(t.cell_ranges), sha(top['res']), sha(leaves)
        tr = tlg_rays(cam)
        _, tst = t.trace_rays(tr)
        assert tst['dropped_rays'] == 0
        got = ref.tlg_generate_contacts2(tr, buf, t.grid, t.top_cells, t.leaf_ranges, t.leaf_pairs)
        hit = got['normal_t'][:, 3] > 0
        g['rays'] = int(len(tr)); g['rays_sha'] = sha(tr)
        g['key_sha'] = sha(contact_key(got)); g['hits'] = int(hit.sum())
        g['sample'] = contact_key(got)[::997].tolist()
        m['tlg'] = g
        out['models'][name] = m
        print(name, 'done', flush=True)
    ref.close()
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', out_path)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'reference_cl.json'))
