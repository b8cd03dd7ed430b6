"""Exploration, part 2: the oracle with ROCm's OpenCL definitions of dot / cross against the reference's ray kernels, and the
Two-Level-Grid / scan kernels that stay in range.  Prints, asserts nothing."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tools')]
import oracle as orc, scene_pack, ref_cl
from ref_cl_probe import cmp_contacts, ndiff

models = sys.argv[1:] or ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
os.chdir(ROOT)
ref = ref_cl.RefCL('strict')
rng = np.random.default_rng(7)
for size in (2, 64, 512):
    v = rng.integers(0, 1024, size=size, dtype=np.uint32)
    got = ref.prefix_sum_one_group(v)
    print('scan size %d: equals inclusive cumsum %s (exclusive %s)' % (size, bool((got == np.cumsum(v, dtype=np.uint32)).all()),
          bool((got[1:] == np.cumsum(v, dtype=np.uint32)[:-1]).all())), got[:6], v[:6], flush=True)
for name in models:
    buf = scene_pack.load_scene(os.path.join('assets', 'scenes', name + '.txt'))
    sc = orc.Scene(buf)
    n = sc.n_tris
    print('== %s n=%d' % (name, n), flush=True)
    for mode in (0, 1):
        ob = orc.BVH(sc, mode)
        cam = sc.demo_camera(320, 240)
        _, st = ob.trace_primary(cam)
        if st.max_stack >= 30:
            print('   skip mode %d: max stack %d' % (mode, st.max_stack)); continue
        rays = orc.generate_rays(cam)
        got = ref.generate_contacts2(rays, ob.nodes, n, buf)
        orc.set_device_arith(True)
        want, _ = ob.trace_rays(rays)
        orc.set_device_arith(False)
        cmp_contacts('mode %d rays, device arith' % mode, got, want)
    # Two-Level Grid
    t = orc.TLG(sc)
    info = t.info
    print('   tlg info', info, flush=True)
    hist = np.bincount(t.top_pairs['value'][t.top_pairs['key'] != 0xFFFFFFFF].astype(np.int64), minlength=n).astype(np.uint32)
    cnt = ref.tlg_prepare_data(buf, t.grid, n)
    print('   prepareData: counts differing (tris 0..n-2) %d, last triangle counter %d (oracle %d)' % ((cnt[:n - 1] != hist[:n - 1]).sum(), cnt[n - 1], hist[n - 1]), flush=True)
    top, leaves = ref.tlg_fill_top_level_cells(t.cell_ranges, t.grid)
    dres = (top['res'] != t.top_cells['res']).any(axis=1)
    print('   fillTopLevelCell: cells %d, resolution differing %d, leaf counts differing %d' % (len(top), dres.sum(),
          (leaves != (t.top_cells['res'].astype(np.uint64).prod(axis=1) & 0xFFFFFFFF)).sum()), flush=True)
    if dres.any():
        k = np.nonzero(dres)[0][:5]
        print('      ', k, top['res'][k], t.top_cells['res'][k], t.cell_ranges[k])
    cam = sc.demo_camera(320, 240)
    rays = orc.generate_rays(cam)
    d = rays['direction'][:, :3]
    ok = (np.isfinite(d) & (d != 0)).all(axis=1)
    rays = rays[ok].copy()
    rays['idx'] = np.arange(len(rays))
    orc.set_device_arith(True)
    want, stt = t.trace_rays(rays)
    orc.set_device_arith(False)
    print('   tlg oracle stats', stt, flush=True)
    if stt['dropped_rays'] == 0:
        got = ref.tlg_generate_contacts2(rays, buf, t.grid, t.top_cells, t.leaf_ranges, t.leaf_pairs)
        cmp_contacts('tlg rays, device arith', got, want)
        want_h, _ = t.trace_rays(rays)
        cmp_contacts('tlg rays, host arith', got, want_h)
ref.close()
