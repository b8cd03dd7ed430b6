"""More GPU parity tests through the C ABI: the ray-buffer entry point, shading, frame tiling, the sort / scan
primitives, edge cases and error behaviour."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import MODELS, ROOT, as_capi_camera, gpu_bvh, oracle_bvh

pytestmark = pytest.mark.gpu


def _rays_capi(rays):
    from openclraytracer_amd import capi
    return rays.view(capi.RAY)


@pytest.mark.parametrize('name,res', [('cube', (64, 48)), ('gardenlight', (100, 75)), ('clumsydragon', (160, 120))])
def test_ray_buffer_entry_equals_primary(gpu_ctx, name, res):
    """generateContacts(CLBuffer& rays, ...) == generateContacts(Camera&) on the camera's own rays (BVHKernels.cl:103-117)"""
    import oracle as orc
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    rays = orc.generate_rays(cam)
    want, _ = ob.trace_rays(rays)
    got = gb.generate_contacts_rays(_rays_capi(rays))
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize('name,res', [('cube', (64, 48)), ('man', (333, 77))])
def test_primary_rays_as_a_ray_buffer(gpu_ctx, name, res):
    """rt_generate_primary_rays: the camera's rays as Ray[] (generateRay, CLStructs.h:95-106), bit for bit the oracle's, and the
    ray-buffer entry on them gives the primary frame's contacts."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    npx = res[0] * res[1]
    d_rays, d_c = gpu_ctx.alloc(npx * 48), gpu_ctx.alloc(npx * 32)
    try:
        gpu_ctx.fill_u32(d_rays, 0xDEADBEEF, npx * 12)
        capi.generate_primary_rays_d(gpu_ctx, as_capi_camera(cam), d_rays)
        got = gpu_ctx.download(d_rays, npx, capi.RAY)
        assert got.tobytes() == orc.generate_rays(cam).tobytes()
        gb.generate_contacts_rays_d(d_rays, npx, d_c)
        want, _ = ob.trace_primary(cam)
        assert gpu_ctx.download(d_c, npx, capi.CONTACT).tobytes() == want.tobytes()
    finally:
        gpu_ctx.free(d_rays); gpu_ctx.free(d_c)


def test_ray_buffer_ragged_random_and_degenerate(gpu_ctx):
    """seeded incoherent rays, a count that is not a multiple of 64, axis-parallel / zero / NaN directions."""
    import oracle as orc
    osc, ob = oracle_bvh('tenryuu')
    _, gb = gpu_bvh(gpu_ctx, 'tenryuu')
    rng = np.random.default_rng(7)
    n = 10007
    rays = np.zeros(n, dtype=orc.RAY)
    rays['idx'] = np.arange(n)
    rays['origin'][:, :3] = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays['direction'][:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays['direction'][0, :3] = (0, 0, 1)
    rays['direction'][1, :3] = (0, 1, 0)
    rays['direction'][2, :3] = (1, 0, 0)
    rays['direction'][3, :3] = (0, 0, 0)
    rays['direction'][4, :3] = (np.nan, 0, 1)
    rays['origin'][5, :3] = (0, 0.5, 0)            # inside the model's bounding box
    want, _ = ob.trace_rays(rays)
    got = gb.generate_contacts_rays(_rays_capi(rays))
    assert got.tobytes() == want.tobytes()
    assert len(gb.generate_contacts_rays(_rays_capi(rays[:0]))) == 0
    one = gb.generate_contacts_rays(_rays_capi(rays[:1]))
    assert one.tobytes() == want[:1].tobytes()


@pytest.mark.parametrize('name', ['cube', 'tenryuu', 'clumsydragon'])
def test_ray_buffer_rays_parallel_to_the_yz_plane_on_box_planes(gpu_ctx, name, monkeypatch):
    """Rays of a ray buffer with direction.x = +-0 keep to the search tree (trace_kernels.hip, ray_root<0>): their 1/d.x is infinite, and
    where the origin's x equals a plane of a box the slab product is 0 * inf = NaN -- which the reference's staged test treats as "no
    constraint on x" ("the first axis") and so do the sign-selected tests of the lane-per-ray and wave-per-ray kernels;
    the four-wide packet kernel hands such rays to the wave-per-ray kernel at the search tree's root.  Origins on vertex x coordinates (= box
    planes), inside and outside the model, both signs of zero, every strategy: the oracle's contacts (reference tree, staged test)."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, ob = oracle_bvh(name)
    sc, _ = gpu_bvh(gpu_ctx, name)
    v, _, _ = osc.triangles() if osc.n_tris <= 20000 else (None, None, None)
    rng = np.random.default_rng(11)
    lo, hi = ob.nodes['lo'][ob.n, :3], ob.nodes['hi'][ob.n, :3]
    if v is not None:
        xs = np.unique(v[:, :, 0].ravel())
    else:
        xs = np.unique(np.concatenate([ob.nodes['lo'][:ob.n, 0], ob.nodes['hi'][:ob.n, 0]]))
    n = 6000
    rays = np.zeros(n, dtype=orc.RAY)
    rays['idx'] = np.arange(n)
    rays['origin'][:, 0] = rng.choice(xs, size=n)
    rays['origin'][:, 1] = rng.uniform(lo[1] - 1, hi[1] + 1, size=n).astype(np.float32)
    rays['origin'][:, 2] = rng.uniform(lo[2] - 3, lo[2] - 0.5, size=n).astype(np.float32)
    rays['origin'][n // 2:, 2] = rng.uniform(lo[2], hi[2], size=n - n // 2).astype(np.float32)        # inside the model's box as well
    ang = rng.uniform(-1.2, 1.2, size=n)
    rays['direction'][:, 0] = np.where(rng.integers(0, 2, size=n) == 0, np.float32(0.0), np.float32(-0.0))
    rays['direction'][:, 1] = np.sin(ang).astype(np.float32)
    rays['direction'][:, 2] = np.cos(ang).astype(np.float32)
    rays['direction'][:50, 1] = 0.0; rays['direction'][:50, 2] = 1.0                                 # ... and x and y both zero: the reference tree
    want, st = ob.trace_rays(rays)
    assert st.hits > 50
    for strategy in ('lane:16', 'lane:1024', 'p4', 'p4s:32'):
        # (the override is read once per process: a child per strategy)
        import subprocess, sys as _sys, tempfile
        with tempfile.TemporaryDirectory(prefix='rtmi_test_') as d:
            np.save(os.path.join(d, 'r.npy'), rays.view(np.uint8)); open(os.path.join(d, 's.bin'), 'wb').write(__import__('conftest').packed_scene(name))
            code = ("import sys, numpy as np; sys.path.insert(0, %r); from openclraytracer_amd import capi; ctx = capi.Context(0); "
                    "sc = capi.DeviceScene(ctx, open(sys.argv[1], 'rb').read()); b = capi.DeviceBVH(ctx, sc).build(); "
                    "np.save(sys.argv[3], b.generate_contacts_rays(np.load(sys.argv[2]).view(capi.RAY)).view(np.uint8))" % ROOT)
            r = subprocess.run([_sys.executable, '-c', code, os.path.join(d, 's.bin'), os.path.join(d, 'r.npy'), os.path.join(d, 'c.npy')],
                               env=dict(os.environ, RTMI_RAYS=strategy), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            got = np.load(os.path.join(d, 'c.npy')).view(capi.CONTACT).reshape(-1)
        assert got.tobytes() == want.tobytes(), strategy


def test_ray_buffer_calibrations_are_keyed_by_what_the_buffer_holds(gpu_ctx):
    """the ray-buffer entry kept ONE calibration per size class, so the strategy found on a camera's rays served the reflections
    of the same size.  A calibration now belongs to (size class, what the buffer holds): the library's generators note what they wrote --
    camera rays, shadow rays, reflections -- and a buffer of the caller's own is known by its address; an entry whose launches come to take
    twice its calibrated time is retired (rt_bvh_rays_calibrations).  Contacts stay the oracle's through all of it."""
    import oracle as orc
    from openclraytracer_amd import capi
    import bench
    ctx = gpu_ctx
    osc, ob = oracle_bvh('cube')                       # (more than half of its pixels are hits: bounces of at least 32 768 rays)
    sc = capi.DeviceScene(ctx, __import__('conftest').packed_scene('cube'))
    gb = capi.DeviceBVH(ctx, sc).build()              # a BVH of this test's own: its calibrations are counted
    cam = osc.demo_camera(640, 360)
    ccam = as_capi_camera(cam)
    npx = 640 * 360
    light, eps = bench.bench_fixture()['light'], bench.bench_fixture()['eps']
    d_r0, d_r1, d_c0, d_c1 = ctx.alloc(npx * 48), ctx.alloc(npx * 48), ctx.alloc(npx * 32), ctx.alloc(npx * 32)
    try:
        assert gb.rays_calibrations() == dict(entries=0, recalibrations=0)
        capi.generate_primary_rays_d(ctx, ccam, d_r0)
        gb.generate_contacts_rays_d(d_r0, npx, d_c0); ctx.synchronize()
        want0, _ = ob.trace_primary(cam)
        assert ctx.download(d_c0, npx, capi.CONTACT).tobytes() == want0.tobytes()
        assert gb.rays_calibrations()['entries'] == 1
        gb.generate_contacts_rays_d(d_r0, npx, d_c0); ctx.synchronize()
        assert gb.rays_calibrations()['entries'] == 1                                   # the same kind and size: no second calibration
        rays0 = orc.generate_rays(cam)
        entries = 1
        for kind in (1, 0):                                                             # reflections, then shadow rays, through the SAME buffer
            n = capi.generate_secondary_rays_d(ctx, d_c0, npx, kind, d_r1, light, eps, d_rays_in=d_r0)
            want_r = orc.generate_secondary_rays(want0, kind, light, eps=eps, rays_in=rays0)
            assert n == len(want_r) and n >= 32768
            gb.generate_contacts_rays_d(d_r1, n, d_c1); ctx.synchronize()
            want_c, _ = ob.trace_rays(want_r)
            assert ctx.download(d_c1, n, capi.CONTACT).tobytes() == want_c.tobytes()
            entries += 1
            assert gb.rays_calibrations()['entries'] == entries, kind                  # its own calibration although size class and address repeat
        # a buffer of the caller's own: known by its address; bytes uploaded over a generated buffer make it the caller's
        # (on a model whose random rays cost many times what rays that leave the scene cost)
        gb.close(); sc.close()
        osc, ob = oracle_bvh('tenryuu')
        sc = capi.DeviceScene(ctx, __import__('conftest').packed_scene('tenryuu'))
        gb = capi.DeviceBVH(ctx, sc).build()
        entries = 0
        rng = np.random.default_rng(5)
        n = 200000
        away = np.zeros(n, dtype=orc.RAY)
        away['idx'] = np.arange(n)
        away['origin'][:, :3] = (0.0, 0.0, -50.0)
        away['direction'][:, :3] = (0.0, 0.0, -1.0)                                     # every ray leaves the scene: the cheapest buffer there is
        ctx.upload(d_r1, away.view(np.uint8))
        gb.generate_contacts_rays_d(d_r1, n, d_c1); ctx.synchronize()
        assert gb.rays_calibrations() == dict(entries=entries + 1, recalibrations=0)
        assert not ctx.download(d_c1, n, capi.CONTACT)['normalAndDistance'][:, 3].any()
        heavy = away.copy()
        heavy['origin'][:, :3] = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        heavy['direction'][:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        ctx.upload(d_r1, heavy.view(np.uint8))                                          # same address, same size class, another workload
        want_h, _ = ob.trace_rays(heavy)
        for _ in range(4):
            gb.generate_contacts_rays_d(d_r1, n, d_c1); ctx.synchronize()
            assert ctx.download(d_c1, n, capi.CONTACT).tobytes() == want_h.tobytes()
        got = gb.rays_calibrations()
        assert got['recalibrations'] == 1 and got['entries'] == entries + 1, got        # retired after two slow launches, calibrated again once
    finally:
        for d_ in (d_r0, d_r1, d_c0, d_c1):
            ctx.free(d_)
        gb.close(); sc.close()


@pytest.mark.parametrize('name', MODELS)
def test_shading_within_tolerance(gpu_ctx, name):
    """device Blinn-Phong vs cpuShadeBlinnPhong restatement: 1e-5 relative on RGB (pow differs in the last ulps),
    alpha and the zero pixels of misses exact."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, ob = oracle_bvh(name)
    sc, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(320, 240)
    contacts, _ = ob.trace_primary(cam)
    want = orc.shade_frame(osc, cam, contacts)
    got = capi.shade_blinn_phong(gpu_ctx, sc, as_capi_camera(cam), contacts.view(capi.CONTACT))
    TOL = 1e-5
    rel = np.abs(got[:, :3] - want[:, :3]) / np.maximum(np.abs(want[:, :3]), 1e-30)
    rel = np.where(got[:, :3] == want[:, :3], 0.0, rel)
    assert float(rel.max()) <= TOL
    assert np.array_equal(got[:, 3], want[:, 3])
    miss = contacts['normal_t'][:, 3] <= 0
    assert (got[miss] == 0).all()
    # the frame variant for a primary contact buffer (contact i = pixel i, no clearing pass): the same RGBA, bit for bit,
    # also into a buffer full of garbage
    npx = 320 * 240
    dc, dr = gpu_ctx.alloc(npx * 32), gpu_ctx.alloc(npx * 16)
    gpu_ctx.upload(dc, contacts.view(capi.CONTACT))
    gpu_ctx.fill_u32(dr, 0xDEADBEEF, npx * 4)
    capi.shade_blinn_phong_frame_d(gpu_ctx, sc, as_capi_camera(cam), dc, dr)
    frame = gpu_ctx.download(dr, npx * 4, np.float32).reshape(npx, 4)
    gpu_ctx.free(dc); gpu_ctx.free(dr)
    assert frame.tobytes() == got.tobytes()


def test_shade_empty_contact_list_clears_frame(gpu_ctx):
    from openclraytracer_amd import capi
    osc, _ = oracle_bvh('cube')
    sc, _ = gpu_bvh(gpu_ctx, 'cube')
    cam = osc.demo_camera(16, 8)
    out = capi.shade_blinn_phong(gpu_ctx, sc, as_capi_camera(cam), np.zeros(0, dtype=capi.CONTACT))
    assert (out == 0).all()


@pytest.mark.parametrize('n_ranks, root_extra', [(2, 0), (3, 0), (8, 0), (2, 1), (4, 1), (8, 2), (3, 5)])
def test_tiled_frame_equals_untiled(gpu_ctx, n_ranks, root_extra):
    """every rank's compact buffers, gathered and untiled, equal the single-GPU frame (contacts bit-exact, RGBA too), for
    plain round robin and for tilings in which rank 0 owns extra tiles per period (RT_TILING)."""
    from openclraytracer_amd import capi
    nranks = capi.tiling(n_ranks, root_extra)
    osc, _ = oracle_bvh('gardenlight')
    sc, gb = gpu_bvh(gpu_ctx, 'gardenlight')
    res = (333, 200)
    cam = as_capi_camera(osc.demo_camera(*res))
    npx = res[0] * res[1]
    full = gb.generate_contacts_primary(cam)
    full_rgba = capi.shade_blinn_phong(gpu_ctx, sc, cam, full)
    slots0 = capi.tiles_owned_pixels(cam, 0, nranks)
    gathered = np.zeros((n_ranks, slots0, 4), dtype=np.float32)
    packed = np.zeros((n_ranks, 3, slots0), dtype=np.float32)
    assert all(capi.tiles_owned_pixels(cam, r, nranks) <= slots0 for r in range(n_ranks))      # rank 0 sizes the gather
    assert sum(capi.tiles_owned_pixels(cam, r, nranks) for r in range(n_ranks)) == 6 * 4 * 4096     # every tile has one owner
    from openclraytracer_amd import frame_tiles as ft
    rank_of, slot_of = capi.tile_slot_map(res[0], res[1], nranks)
    for r in range(n_ranks):
        slots = capi.tiles_owned_pixels(cam, r, nranks)
        dc, dr = gpu_ctx.alloc(max(slots, 1) * 32), gpu_ctx.alloc(max(slots, 1) * 16)
        gpu_ctx.fill_u32(dc, 0xDEADBEEF, max(slots, 1) * 8)
        gb.generate_contacts_primary_tiled_d(cam, r, nranks, dc)
        capi.shade_blinn_phong_tiled_d(gpu_ctx, sc, cam, r, nranks, dc, dr)
        compact = gpu_ctx.download(dc, slots, capi.CONTACT)
        rgba = gpu_ctx.download(dr, slots * 4, np.float32).reshape(slots, 4)
        mine = rank_of == r
        assert compact[slot_of[mine]].tobytes() == full[mine].tobytes()
        other = np.ones(slots, dtype=bool)
        other[slot_of[mine]] = False
        assert (compact[other]['normalAndDistance'][:, 3] == 0).all()          # slots outside the frame
        gathered[r, :slots] = rgba
        # the 12-byte wire format of the same buffer: three planes, hit flag in the sign bit of red
        dp = gpu_ctx.alloc(slots0 * 12)
        gpu_ctx.fill_u32(dp, 0xDEADBEEF, slots0 * 3)
        capi.shade_blinn_phong_tiled_packed_d(gpu_ctx, sc, cam, r, nranks, dc, dp, slots0)
        planes = gpu_ctx.download(dp, slots0 * 3, np.float32).reshape(3, slots0)
        assert planes[:, :slots].tobytes() == ft.pack_planes(rgba).tobytes()
        packed[r, :, :slots] = planes[:, :slots]
        gpu_ctx.free(dc); gpu_ctx.free(dr); gpu_ctx.free(dp)
    dg, df = gpu_ctx.alloc(gathered.nbytes), gpu_ctx.alloc(npx * 16)
    gpu_ctx.upload(dg, gathered)
    capi.untile_rgba_d(gpu_ctx, cam, nranks, dg, slots0, df)
    frame = gpu_ctx.download(df, npx * 4, np.float32).reshape(npx, 4)
    assert frame.tobytes() == full_rgba.tobytes()
    gpu_ctx.upload(dg, packed)
    gpu_ctx.fill_u32(df, 0xDEADBEEF, npx * 4)
    capi.untile_rgba_packed_d(gpu_ctx, cam, nranks, dg, slots0, df)
    frame = gpu_ctx.download(df, npx * 4, np.float32).reshape(npx, 4)
    gpu_ctx.free(dg); gpu_ctx.free(df)
    assert frame.tobytes() == full_rgba.tobytes()
    assert ft.untile_host_packed(packed, res[0], res[1], nranks).tobytes() == full_rgba.tobytes()      # the host statement agrees
    assert ft.untile_host(gathered, res[0], res[1], nranks).tobytes() == full_rgba.tobytes()


def test_sort_and_scan_primitives(gpu_ctx):
    """same comparator network => same permutation of equal keys as the reference's bitonic sort; inclusive scan."""
    import oracle as orc
    from openclraytracer_amd import capi
    kat = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'kat.json')))
    rng = np.random.default_rng(1234)
    for n in [2, 64, 1024, 65536]:
        a = rng.integers(0, 1 << 10, size=n, dtype=np.uint32)
        s = gpu_ctx.scan_inclusive(a)
        assert hashlib.sha256(s.tobytes()).hexdigest() == kat['scan_%d' % n]
        pairs = np.zeros(n, dtype=capi.PAIR)
        pairs['key'] = rng.integers(0, 16, size=n, dtype=np.uint32)
        pairs['value'] = np.arange(n, dtype=np.uint32)
        out = gpu_ctx.sort_pairs_bitonic(pairs)
        assert hashlib.sha256(out.tobytes()).hexdigest() == kat['sort_%d' % n]
    # sizes around the LDS tile and the multi-level scan, random full-range keys
    for n in [4096, 8192, 1 << 18]:
        pairs = np.zeros(n, dtype=capi.PAIR)
        pairs['key'] = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        pairs['value'] = rng.permutation(n).astype(np.uint32)
        assert gpu_ctx.sort_pairs_bitonic(pairs).tobytes() == orc.bitonic_sort(pairs.view(orc.PAIR)).tobytes()
    # up to nine levels through HBM per merge (eight fused in one pass + one): few distinct keys stress the tie order
    for n in [1 << 20, 1 << 21]:
        pairs = np.zeros(n, dtype=capi.PAIR)
        pairs['key'] = rng.integers(0, 64, size=n, dtype=np.uint32)
        pairs['key'][rng.integers(0, n, size=n // 7)] = 0xFFFFFFFF             # padding-like entries anywhere
        pairs['value'] = np.arange(n, dtype=np.uint32)
        assert gpu_ctx.sort_pairs_bitonic(pairs).tobytes() == orc.bitonic_sort(pairs.view(orc.PAIR)).tobytes()
    for n in [1, 63, 2049, 5000000]:
        a = rng.integers(0, 100, size=n, dtype=np.uint32)
        assert np.array_equal(gpu_ctx.scan_inclusive(a), np.cumsum(a, dtype=np.uint64).astype(np.uint32))
    assert len(gpu_ctx.scan_inclusive(np.zeros(0, dtype=np.uint32))) == 0


def test_radix_sort_is_a_stable_sort(gpu_ctx):
    """rt_sort_pairs(.., RT_SORT_RADIX): the pairs in ascending key order, equal keys in input order (numpy's stable argsort), for counts
    around every boundary of the implementation (one LDS tile of 4 096, partial last tiles, one wave's 1 024, thousands of tiles), keys
    with few distinct values (every digit run long), full-range keys, keys that differ in one byte only, and already sorted / reversed
    input; RT_SORT_BITONIC through the same entry is the network."""
    import oracle as orc
    from openclraytracer_amd import capi
    rng = np.random.default_rng(20261005)

    def check(pairs):
        got = gpu_ctx.sort_pairs(pairs, 1)
        order = np.argsort(pairs['key'], kind='stable')
        assert got.tobytes() == pairs[order].tobytes(), len(pairs)

    for n in [0, 1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 5121, 8191, 12289, 65535, 65536, 65537, 100003, 262144, 262145, (1 << 20) + 17, 3000001]:
        pairs = np.zeros(n, dtype=capi.PAIR)
        pairs['value'] = np.arange(n, dtype=np.uint32)
        pairs['key'] = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        check(pairs)
        pairs['key'] = rng.integers(0, 5, size=n, dtype=np.uint32)                     # long runs of equal keys: stability
        check(pairs)
        pairs['key'] = rng.integers(0, 256, size=n, dtype=np.uint32) << 16             # one byte differs
        check(pairs)
    n = 50000
    pairs = np.zeros(n, dtype=capi.PAIR)
    pairs['value'] = rng.permutation(n).astype(np.uint32)
    pairs['key'] = np.arange(n, dtype=np.uint32) * 85899                               # sorted
    check(pairs)
    pairs['key'] = pairs['key'][::-1].copy()                                            # reversed
    check(pairs)
    pairs['key'] = 0xFFFFFFFF                                                           # all equal, all bits set
    check(pairs)
    p2 = np.zeros(4096, dtype=capi.PAIR)
    p2['key'] = rng.integers(0, 16, size=4096, dtype=np.uint32); p2['value'] = np.arange(4096, dtype=np.uint32)
    assert gpu_ctx.sort_pairs(p2, 0).tobytes() == orc.bitonic_sort(p2.view(orc.PAIR)).tobytes()
    with pytest.raises(capi.RtError, match='unknown sort algorithm'):
        gpu_ctx.sort_pairs(p2, 5)


@pytest.mark.parametrize('name', MODELS)
def test_fast_mode_keeps_hits_and_distances(gpu_ctx, name):
    """RT_MODE_FAST (the reference tree's Morton pairs through the stable radix sort instead of the reference's network): sorted by code
    like FIXED's pairs, equal codes in leaf order; every node still holds the union of its leaves; contacts: hit mask and t bit for bit
    FIXED's (= the oracle's), normal / material different only where another triangle is hit at exactly the same t."""
    from openclraytracer_amd import capi
    osc, ob = oracle_bvh(name)
    sc, fixed = gpu_bvh(gpu_ctx, name)
    fast = capi.DeviceBVH(gpu_ctx, sc, mode=2).build()
    try:
        pf, px = fast.read_morton_pairs()[:ob.n], fixed.read_morton_pairs()[:ob.n]
        assert np.array_equal(pf['key'], px['key'])                                              # the same codes in the same places
        order = np.lexsort((pf['value'], pf['key']))
        assert np.array_equal(order, np.arange(ob.n))                                            # ... equal codes by leaf index (stable)
        assert sorted(pf['value'].tolist()) == list(range(ob.n))
        for res in [(320, 240), (1920, 1080)]:
            cam = as_capi_camera(osc.demo_camera(*res))
            a, b = fast.generate_contacts_primary(cam), fixed.generate_contacts_primary(cam)
            assert a['pixelIndex'].tobytes() == b['pixelIndex'].tobytes()
            assert a['normalAndDistance'][:, 3].tobytes() == b['normalAndDistance'][:, 3].tobytes()   # hit mask and t
            differ = np.nonzero((a['materialIndex'] != b['materialIndex']) | (a['normalAndDistance'][:, :3] != b['normalAndDistance'][:, :3]).any(axis=1))[0]
            assert len(differ) < 0.002 * len(a) + 8, (name, res, len(differ))
            if len(differ):
                # such a pixel's ray hits two triangles at exactly the same t: the oracle's brute-force list of hits says so
                rays = __import__('oracle').generate_rays(osc.demo_camera(*res))[differ[:16]]
                v, _, _ = osc.triangles() if ob.n <= 20000 else (None, None, None)
                if v is not None:
                    for k, r in enumerate(rays):
                        ts = _all_hit_distances(v, r)
                        t = a['normalAndDistance'][differ[k], 3]
                        assert (ts == t).sum() >= 2, (name, int(differ[k]))
        rays = __import__('oracle').generate_secondary_rays(fixed.generate_contacts_primary(as_capi_camera(osc.demo_camera(200, 150))).view(__import__('oracle').CONTACT), 1,
                                                           eps=1e-3, cam=osc.demo_camera(200, 150))
        if len(rays):
            a, b = fast.generate_contacts_rays(_rays_capi(rays)), fixed.generate_contacts_rays(_rays_capi(rays))
            assert a['normalAndDistance'][:, 3].tobytes() == b['normalAndDistance'][:, 3].tobytes()
    finally:
        fast.close()


def _all_hit_distances(verts, ray):
    """triangleIntersect (Triangle.h:66-107) of one ray against every triangle, in numpy fp32 with the reference's operation order; -> t"""
    f = np.float32
    o, d = ray['origin'][:3].astype(f), ray['direction'][:3].astype(f)
    v0, v1, v2 = verts[:, 0, :3], verts[:, 1, :3], verts[:, 2, :3]
    e1, e2 = v1 - v0, v2 - v0

    def cross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2], a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)

    def dot(a, b):
        return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]
    with np.errstate(all='ignore'):
        pvec = cross(np.broadcast_to(d, e2.shape), e2)
        det = dot(e1, pvec)
        inv = f(1.0) / det
        tvec = o - v0
        u = dot(tvec, pvec) * inv
        qvec = cross(tvec, e1)
        v = dot(np.broadcast_to(d, qvec.shape), qvec) * inv
        t = dot(e2, qvec) * inv
        ok = (np.abs(det) >= np.finfo(f).eps) & (u < 1) & (u >= 0) & (u + v <= 1) & (v >= 0) & (t > 0)
    return t[ok]


def test_error_behaviour(gpu_ctx):
    """0/1 results with a message, nothing thrown across the boundary (Errata convention)."""
    from openclraytracer_amd import capi
    with pytest.raises(capi.RtError, match='smaller than a SceneHeader'):
        capi.DeviceScene(gpu_ctx, b'\0' * 10)
    bad = bytearray(packed := __import__('conftest').packed_scene('cube'))
    bad[56:64] = (99).to_bytes(8, 'little')                     # totalNumberOfTriangles lies
    with pytest.raises(capi.RtError, match='totalNumberOfTriangles'):
        capi.DeviceScene(gpu_ctx, bytes(bad))
    sc = capi.DeviceScene(gpu_ctx, packed)
    with pytest.raises(capi.RtError, match='unknown parity mode'):
        capi.DeviceBVH(gpu_ctx, sc, mode=7)
    b = capi.DeviceBVH(gpu_ctx, sc)
    with pytest.raises(capi.RtError, match='initializeFrame must precede construct'):
        b.construct()
    cam = np.zeros(1, dtype=capi.CAMERA)
    cam['resX'], cam['resY'] = 8, 8
    d = gpu_ctx.alloc(64 * 32)
    with pytest.raises(capi.RtError, match='construct must precede generateContacts'):
        b.generate_contacts_primary_d(cam, d)
    b.build()
    cam['resX'] = 0
    with pytest.raises(capi.RtError, match='resolution'):
        b.generate_contacts_primary_d(cam, d)
    with pytest.raises(capi.RtError, match='power of two'):
        gpu_ctx.check(gpu_ctx.lib.rt_sort_pairs_bitonic(gpu_ctx.h, d, 12))
    gpu_ctx.free(d)
    b.close(); sc.close()


@pytest.mark.parametrize('name,mode', [('cube', 0), ('tenryuu', 0), ('tenryuu', 2), ('gardenlight', 1)])
def test_rebuilds_replay_a_build_graph(gpu_ctx, name, mode):
    """A scene rebuilt every frame (OpenCLRayTracerDemo.cpp:230-234): the first build of an allocation is enqueued as it is, the second
    is captured, later ones replay the graph -- every one of them leaves the oracle's node bytes (FIXED / LITERAL_ZERO) and the same
    contacts, also with traces and another scene's builds in between."""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc = orc.Scene(packed_scene(name))
    ob = orc.BVH(osc, mode if mode != 2 else 0)
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))
    gb = capi.DeviceBVH(gpu_ctx, sc, mode=mode)
    cam = as_capi_camera(osc.demo_camera(200, 150))
    first = None
    for k in range(6):
        gb.build()
        if mode != 2:
            assert gb.read_nodes().tobytes() == ob.nodes.tobytes(), k
            assert gb.read_morton_pairs().tobytes() == ob.pairs.tobytes(), k
        frame = gb.generate_contacts_primary(cam).tobytes()
        first = first or frame
        assert frame == first, k
        if k == 3:
            _, other = gpu_bvh(gpu_ctx, 'man')
            other.build()
    assert gb.graph_builds() >= 4
    gb.close(); sc.close()


def test_frame_graph_policy_state_through_the_abi():
    """The counters behind a graph's block start order, read through rt_frame_graph_state instead of inferred from timings: the order is
    derived behind the first launch, then every 32 frames, and the interval doubles with every refresh while the camera stands still (64,
    128, ... 1 024) and is 32 again once the camera moves; a graph of kernels only reports that it is issued as plain launches."""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene('man'))
        bvh = capi.DeviceBVH(ctx, sc).build()
        osc = orc.Scene(packed_scene('man'))
        cam = as_capi_camera(osc.demo_camera(640, 360))
        npx = 640 * 360
        dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
        g = capi.FrameGraph(bvh, sc, cam, dc, dr)
        st = g.state()
        if st['four_wide'] and st['order_interval']:
            assert st['order_interval'] == 32 and not st['order_sorted'] and not st['moved']
            g.launch()
            assert g.state()['order_sorted'] == 1 and g.state()['order_interval'] == 32 and g.state()['order_age'] == 0
            seen = []
            for _ in range(32 + 64 + 128 + 4):
                g.launch()
                seen.append(g.state()['order_interval'])
            assert seen[30] == 32 and seen[31] == 64 and seen[31 + 63] == 64 and seen[31 + 64] == 128 and seen[31 + 64 + 128] == 256
            cam2 = cam.copy(); cam2['viewTransform'][0][3] += 0.01
            g.set_camera(cam2)
            assert g.state()['order_interval'] == 32 and g.state()['moved'] == 1
            g.launch(); ctx.synchronize()
            assert g.state()['direct'] in (1, 2, 3) or g.state()['eager'] in (0, 1)
        g.close(); ctx.free(dc); ctx.free(dr); bvh.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name', ['tenryuu', 'clumsydragon'])
def test_fused_frame_colours_at_full_hd(name):
    """The colour path of bench.py's timed region -- a captured frame whose trace kernels shade what they store (four-wide kernel: Tenryuu;
    lane-per-ray + wave-per-ray: ClumsyDragon) -- against the oracle's cpuShadeBlinnPhong restatement at 1920x1080, every pixel: RGB within
    1e-5 relative, alpha and the cleared pixels exact, contacts bit for bit.  (Round 3 compared colours at 320x240 and the fused path only
    through graph == eager.)"""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc, ob = oracle_bvh(name)
    cam = osc.demo_camera(1920, 1080)
    want_c, _ = ob.trace_primary(cam)
    want = orc.shade_frame(osc, cam, want_c)
    ctx = capi.Context(0)
    try:
        ctx.set_frames_in_flight(5)
        sc = capi.DeviceScene(ctx, packed_scene(name))
        bvh = capi.DeviceBVH(ctx, sc).build()
        npx = 1920 * 1080
        dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
        g = capi.FrameGraph(bvh, sc, as_capi_camera(cam), dc, dr)
        ctx.fill_u32(dr, 0x7FC00000, npx * 4); ctx.fill_u32(dc, 0xDEADBEEF, npx * 8)
        g.launch(); g.launch(); ctx.synchronize()
        assert ctx.download(dc, npx, capi.CONTACT).tobytes() == want_c.tobytes()
        got = ctx.download(dr, npx * 4, np.float32).reshape(npx, 4)
        assert np.array_equal(got[:, 3], want[:, 3])
        err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
        assert float(np.where(got == want, 0.0, err).max()) <= 1e-5
        g.close(); ctx.free(dc); ctx.free(dr); bvh.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name, knobs', [('cube', {}), ('tenryuu', {}), ('clumsydragon', {}), ('man', {'RTMI_PACKET': '4', 'RTMI_DEFER_BUDGET': '1048576'}),
                                         ('man', {'RTMI_PACKET': '1', 'RTMI_DEFER_BUDGET': '256'})])
def test_render_frame_is_generate_contacts_plus_shade(name, knobs, monkeypatch):
    """rt_render_frame -- one eager frame whose trace kernels shade what they store (the frame of a scene that is rebuilt every frame: a frame
    graph does not survive construct()) -- against the oracle: contacts bit for bit, colours within 1e-5, alpha and cleared pixels exact.  On a
    fresh build, on a rebuilt one, and with the camera on a y and on a z plane of leaf boxes, where the four-wide kernel hands rays with a zero
    direction component to the wave-per-ray kernel, which then shades them (forced strategies: the four-wide kernel, and the binary packet
    kernel, which leaves the colours to k_shade)."""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import plane_rays as pr
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    osc, ob = oracle_bvh(name)
    res = (320, 200)
    cams = [osc.demo_camera(*res)]
    rng = np.random.default_rng(5)
    for axis in (1, 2):
        value = 0.0
        while value == 0.0:
            value = ob.nodes['hi'][int(rng.integers(0, ob.n)), axis]
        cams.append(pr.plane_camera(osc, res, axis, value))
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        bvh = capi.DeviceBVH(ctx, sc).build()
        npx = res[0] * res[1]
        dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
        for round_ in range(2):
            for k, cam in enumerate(cams):
                want_c, _ = ob.trace_primary(cam)
                want = orc.shade_frame(osc, cam, want_c)
                ctx.fill_u32(dr, 0x7FC00000, npx * 4); ctx.fill_u32(dc, 0xDEADBEEF, npx * 8)
                capi.render_frame_d(bvh, sc, as_capi_camera(cam), dc, dr)
                ctx.synchronize()
                assert ctx.download(dc, npx, capi.CONTACT).tobytes() == want_c.tobytes(), (name, round_, k)
                got = ctx.download(dr, npx * 4, np.float32).reshape(npx, 4)
                assert np.array_equal(got[:, 3], want[:, 3]), (name, round_, k)
                err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
                assert float(np.where(got == want, 0.0, err).max()) <= 1e-5, (name, round_, k)
            bvh.build()                                    # the second round on a rebuilt BVH
        ctx.free(dc); ctx.free(dr); bvh.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name', ['man', 'clumsydragon'])
def test_render_frame_tiled_layouts(name):
    """rt_render_frame with the tiled layouts a multi-GPU frame uses (compact contacts of a rank's 64x64 tiles; colours as compact float4 or as
    three planes of slots_per_rank floats): the contacts of rt_bvh_generate_contacts_primary_tiled bit for bit, the colours of
    rt_shade_blinn_phong_tiled / _tiled_packed within 1e-5 (the fused path evaluates the same function), for ranks 0 and 2 of 3."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc, _ = oracle_bvh(name)
    res = (400, 240)
    cam = as_capi_camera(osc.demo_camera(*res))
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        bvh = capi.DeviceBVH(ctx, sc).build()
        nranks = 3
        for rank in (0, 2):
            owned = capi.tiles_owned_pixels(cam, rank, nranks)
            slots = owned + 128
            dc, dc2 = ctx.alloc(owned * 32), ctx.alloc(owned * 32)
            dr, dr2 = ctx.alloc(owned * 16), ctx.alloc(owned * 16)
            dp, dp2 = ctx.alloc(slots * 12), ctx.alloc(slots * 12)
            # the two calls
            bvh.generate_contacts_primary_tiled_d(cam, rank, nranks, dc)
            capi.shade_blinn_phong_tiled_d(ctx, sc, cam, rank, nranks, dc, dr)
            ctx.fill_u32(dp, 0x7FC00000, slots * 3)
            capi.shade_blinn_phong_tiled_packed_d(ctx, sc, cam, rank, nranks, dc, dp, slots)
            ctx.synchronize()
            want_c = ctx.download(dc, owned, capi.CONTACT).tobytes()
            want_r = ctx.download(dr, owned * 4, np.float32)
            want_p = ctx.download(dp, slots * 3, np.float32).reshape(3, slots)[:, :owned]
            # one sequence
            for layout, d_out, want in ((capi.FRAME_TILED, dr2, want_r), (capi.FRAME_TILED_PACKED, dp2, want_p)):
                ctx.fill_u32(dc2, 0xDEADBEEF, owned * 8)
                ctx.fill_u32(d_out, 0x7FC00000, owned * 4 if layout == capi.FRAME_TILED else slots * 3)
                capi.render_frame_d(bvh, sc, cam, dc2, d_out, rank=rank, nranks=nranks, layout=layout, slots_per_rank=slots if layout == capi.FRAME_TILED_PACKED else 0)
                ctx.synchronize()
                assert ctx.download(dc2, owned, capi.CONTACT).tobytes() == want_c, (name, rank, layout)
                got = ctx.download(d_out, owned * 4, np.float32) if layout == capi.FRAME_TILED else ctx.download(d_out, slots * 3, np.float32).reshape(3, slots)[:, :owned]
                err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
                assert float(np.where(got == want, 0.0, err).max()) <= 1e-5, (name, rank, layout)
            for d in (dc, dc2, dr, dr2, dp, dp2):
                ctx.free(d)
        bvh.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name', ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon'])
def test_search_tree_equals_its_restatement(gpu_ctx, name):
    """The search tree the library builds (rt_bvh_read_search_nodes) against its restatement with the oracle's building blocks
    (tests/golden/make_golden_search.py: the keys of k_morton_leaves -- box centres in the scene header's box, the key bits allocated by the
    box's shape (key_layout) --, the reference's network, radix tree, plain unions): every node byte.  Pins the fixture bench.py takes its
    `walked` bytes from, and the key layout on boxes of three shapes (10/10/10, 9/12/9, 11/11/8 bits)."""
    import sys
    import oracle as orc
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_golden_search as mgs
    from conftest import packed_scene
    buf = packed_scene(name)
    osc, ob = oracle_bvh(name)
    hdr = np.frombuffer(buf, dtype=np.float32, count=24, offset=0)
    want, depth = mgs.search_tree(orc, ob, (hdr[16:19].copy(), hdr[20:23].copy()))
    _, gb = gpu_bvh(gpu_ctx, name)
    got = gb.read_search_nodes()
    assert gb.search_tree()['depth'] == depth
    n = ob.n
    assert got[:n].tobytes() == want[:n].tobytes()                                      # the leaves
    assert np.array_equal(got['data'][n:, 1:3], want['data'][n:, 1:3]), 'children differ: another key or another order of equal keys'
    assert np.array_equal(got['boundsMin'][n:, :3], want['lo'][n:, :3]) and np.array_equal(got['boundsMax'][n:, :3], want['hi'][n:, :3])


def test_rebuild_is_deterministic(gpu_ctx):
    """initializeFrame + construct can be repeated (per-frame rebuild, OpenCLRayTracerDemo.cpp:230-234)."""
    _, gb = gpu_bvh(gpu_ctx, 'man')
    a = gb.read_nodes().tobytes()
    for _ in range(3):
        gb.build()
        assert gb.read_nodes().tobytes() == a


def test_cpp_host_mirror_end_to_end(gpu_ctx, tmp_path):
    """the headless Demo written against the reference's class names (Scene, BVHManager, CLBuffer, Errata):
    its contacts are the oracle's, bit for bit."""
    import subprocess
    osc, ob = oracle_bvh('gardenlight')
    demo = os.path.join(ROOT, 'apps', 'bin', 'rt_demo')
    assert os.path.exists(demo), 'apps/bin/rt_demo is not built: run __graft_entry__.build()'
    raw, ppm = str(tmp_path / 'c.bin'), str(tmp_path / 'o.ppm')
    r = subprocess.run([demo, '-winH', '120', '-winW', '160', '-accStruct', 'BVH', '-headersPath', 'unused', '-scene',
                        os.path.join('assets', 'scenes', 'gardenlight.txt'), '--frames', '2', '--contacts', raw, '--out', ppm],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want, _ = ob.trace_primary(osc.demo_camera(160, 120))
    assert open(raw, 'rb').read() == want.tobytes()
    assert open(ppm, 'rb').read().startswith(b'P6\n160 120\n255\n')
    # float frame (PFM, rows bottom to top): the shaded RGB itself, within 1e-5 relative of the oracle's host shading
    import oracle as orc
    pfm = str(tmp_path / 'o.pfm')
    r = subprocess.run([demo, '-winH', '120', '-winW', '160', '-accStruct', 'BVH', '-headersPath', 'unused', '-scene',
                        os.path.join('assets', 'scenes', 'gardenlight.txt'), '--out', pfm], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = open(pfm, 'rb').read()
    head = b'PF\n160 120\n-1.0\n'
    assert blob.startswith(head) and len(blob) == len(head) + 160 * 120 * 12
    rgb = np.frombuffer(blob, dtype='<f4', offset=len(head)).reshape(120, 160, 3)[::-1]
    cam = osc.demo_camera(160, 120)
    want_rgba = orc.shade_frame(osc, cam, want).reshape(120, 160, 4)
    err = np.abs(rgb - want_rgba[:, :, :3]) / np.maximum(np.abs(want_rgba[:, :, :3]), 1e-30)
    assert float(np.where(rgb == want_rgba[:, :, :3], 0.0, err).max()) <= 1e-5
    r = subprocess.run([demo, '-winH', '8', '-winW', '8', '-accStruct', 'OCTREE', '-headersPath', 'x', '-scene', 'x'],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and 'Unsupported acceleration structure' in r.stdout
    # -accStruct GRID: TwoLevelGridManager behind the same abstract interface (OpenCLRayTracerDemo.cpp:221-224)
    r = subprocess.run([demo, '-winH', '120', '-winW', '160', '-accStruct', 'GRID', '-headersPath', 'unused', '-scene',
                        os.path.join('assets', 'scenes', 'gardenlight.txt'), '--frames', '2', '--contacts', raw],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want, _ = orc.TLG(osc).trace_primary(osc.demo_camera(160, 120))
    assert open(raw, 'rb').read() == want.tobytes()


@pytest.mark.parametrize('packet,budget,list_max', [('1', '64', '3000'), ('1', '16', '70'), ('0', '32', '1000'),
                                                     ('0', '1048576', '0'), ('1', '1048576', '0'), ('4', '1048576', '0')])
def test_forced_trace_strategies_and_full_deferred_list(packet, budget, list_max, monkeypatch):
    """every (kernel, budget) strategy the first-frame calibration can pick gives the reference's contacts, also when the
    deferred-ray list is far too small for what the kernels want to hand over (list-full path: rays keep walking in
    place, reserved-but-unwritten slots are void records)."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_POISON', '7')        # new list / key buffers look like recycled memory, not zeros
    monkeypatch.setenv('RTMI_PACKET', packet)
    monkeypatch.setenv('RTMI_DEFER_BUDGET', budget)
    if list_max != '0':
        monkeypatch.setenv('RTMI_LONG_LIST_MAX', list_max)
    ctx = capi.Context(0)
    try:
        # the lane-per-ray kernel may only run out of list where no ray outgrows its 24-entry LDS stack (a forced
        # hand-off with a full list is a reported error, not a silent wrong frame): man's tree is 187 levels deep
        cases = (('tenryuu', (320, 240)),) if (packet == '0' and list_max != '0') else (('tenryuu', (320, 240)), ('man', (200, 150)))
        for name, res in cases:
            osc, ob = oracle_bvh(name)
            sc = capi.DeviceScene(ctx, packed_scene(name))
            gb = capi.DeviceBVH(ctx, sc).build()
            cam = osc.demo_camera(*res)
            want, _ = ob.trace_primary(cam)
            for _ in range(2):                                   # second frame reuses stale list contents
                got = gb.generate_contacts_primary(as_capi_camera(cam))
                assert got.tobytes() == want.tobytes(), (name, packet, budget, list_max)
            gb.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('knobs', [dict(RTMI_DEFER_BUDGET='4'), dict(RTMI_DEFER_BUDGET='64'),
                                   dict(RTMI_DEFER_BUDGET='4', RTMI_TASK_CAP='1'),            # queues of one entry: most entries stay with their publisher
                                   dict(RTMI_DEFER_BUDGET='4', RTMI_LONG_LIST_MAX='40'),      # list of shared blocks full: later blocks stay whole
                                   dict(RTMI_DEFER_BUDGET='8', RTMI_TASK_HELPERS='1'),        # one helper block: three takers and the watching wave
                                   dict(RTMI_DEFER_BUDGET='4', RTMI_TASK_CAP='0'),            # sharing switched off: the plain kernel
                                   dict(RTMI_DEFER_BUDGET='1048576', RTMI_NO_OCTANT_RECORDS='1'),   # the instances that walk the general camera-relative copy
                                   dict(RTMI_DEFER_BUDGET='16', RTMI_NO_OCTANT_RECORDS='1')],
                         ids=['budget4', 'budget64', 'queues-of-one', 'block-list-full', 'one-helper-block', 'no-queues',
                              'general-records', 'general-records-shared'])
def test_four_wide_packets_sharing_sub_trees(knobs, monkeypatch):
    """k_trace_packet4 with a step budget: waves publish their waiting (node, lane mask) entries as tasks, other waves walk them,
    the parts of a block meet in the (t, position) keys and the helper waves write those blocks' contacts.  Whatever the
    budget, the queue capacity and the number of helpers, the contacts are the reference's, frame after frame (the queues and
    keys of the previous frame are stale memory by then)."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_POISON', '7')
    monkeypatch.setenv('RTMI_PACKET', '4')
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    ctx = capi.Context(0)
    try:
        for name, res in (('tenryuu', (320, 240)), ('man', (200, 150)), ('cube', (64, 48))):
            osc, ob = oracle_bvh(name)
            sc = capi.DeviceScene(ctx, packed_scene(name))
            gb = capi.DeviceBVH(ctx, sc).build()
            cam = osc.demo_camera(*res)
            want, _ = ob.trace_primary(cam)
            for _ in range(3):
                got = gb.generate_contacts_primary(as_capi_camera(cam))
                assert got.tobytes() == want.tobytes(), (name, knobs)
            c = ctx.last_trace_counters()
            if knobs.get('RTMI_TASK_CAP') == '0' or knobs['RTMI_DEFER_BUDGET'] == '1048576':
                assert c['shared_blocks'] == 0 and c['shared_tasks'] == 0
            elif name != 'cube' and knobs['RTMI_DEFER_BUDGET'] == '4':
                assert c['shared_blocks'] > 0 and c['shared_tasks'] > 0, (name, knobs, c)
                if knobs.get('RTMI_LONG_LIST_MAX'):
                    assert c['shared_blocks'] >= 40        # the counter runs on, the list holds the first 40
            # a tiled (compact) frame of two ranks takes the same path: shared blocks' contacts go to their compact slots
            ccam = as_capi_camera(cam)
            nranks = capi.tiling(2, 0)
            rank_of, slot_of = capi.tile_slot_map(res[0], res[1], nranks)
            for rank in range(2):
                slots = capi.tiles_owned_pixels(ccam, rank, nranks)
                dc = ctx.alloc(max(slots, 1) * 32)
                ctx.fill_u32(dc, 0xDEADBEEF, max(slots, 1) * 8)
                gb.generate_contacts_primary_tiled_d(ccam, rank, nranks, dc)
                compact = ctx.download(dc, slots, capi.CONTACT)
                mine = rank_of == rank
                assert compact[slot_of[mine]].tobytes() == want.reshape(-1)[mine].tobytes(), (name, knobs, rank)
                ctx.free(dc)
            gb.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name,res', [('tenryuu', (640, 360)), ('man', (320, 240))])
def test_rebuilt_bvh_keeps_its_strategy_and_contacts(gpu_ctx, name, res):
    """Dynamic scenes rebuild per frame (OpenCLRayTracerDemo.cpp:230-234): initializeFrame + construct + generateContacts.  The
    strategy calibrated on the first frame is a property of how the model fills the frame and is kept across rebuilds of the
    same size (one calibration, not one per frame); the contacts are the oracle's after every rebuild; a frame graph made
    before a rebuild refuses to replay."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc, ob = oracle_bvh(name)
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))            # a BVH of its own: the shared one must not be rebuilt under other tests
    gb = capi.DeviceBVH(gpu_ctx, sc).build()
    cam = osc.demo_camera(*res)
    want, _ = ob.trace_primary(cam)
    ccam = as_capi_camera(cam)
    assert gb.generate_contacts_primary(ccam).tobytes() == want.tobytes()
    first = gb.trace_strategy()
    assert first['budget'] != 0                                  # calibrated
    for _ in range(3):
        gb.build()                                               # initializeFrame + construct
        assert gb.generate_contacts_primary(ccam).tobytes() == want.tobytes()
        assert gb.trace_strategy() == first
    npx = res[0] * res[1]
    dc, dr = gpu_ctx.alloc(npx * 32), gpu_ctx.alloc(npx * 16)
    g = capi.FrameGraph(gb, sc, ccam, dc, dr)
    g.launch()
    assert gpu_ctx.download(dc, npx, capi.CONTACT).tobytes() == want.tobytes()
    gb.build()
    with pytest.raises(RuntimeError, match='stale'):
        g.launch()
    g.close(); gpu_ctx.free(dc); gpu_ctx.free(dr)
    assert gb.generate_contacts_primary(ccam).tobytes() == want.tobytes()
    gb.close(); sc.close()


@pytest.mark.parametrize('batch', ['1', '2', '4'])
def test_wave_per_ray_kernel_with_one_two_and_four_rays_per_wave(batch, monkeypatch):
    """The wave-per-ray kernel walks 1, 2 or 4 deferred rays per wave (chosen from the number of deferred rays per resident
    wave; forced here with RTMI_LONG_BATCH): pool entries carry the ray's slot, hits meet in an LDS atomic min on
    (t, position).  Small hand-over budgets send most rays through it -- whole rays from the lane-per-ray kernel, partial
    rays (sub-trees) from the packet kernel -- and the contacts are the oracle's for every batch size, primary rays and
    ray buffers alike."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import oracle as orc
    monkeypatch.setenv('RTMI_LONG_BATCH', batch)
    monkeypatch.setenv('RTMI_POISON', '7')
    for packet, budget in (('0', '8'), ('1', '64')):
        monkeypatch.setenv('RTMI_PACKET', packet)
        monkeypatch.setenv('RTMI_DEFER_BUDGET', budget)
        ctx = capi.Context(0)
        try:
            for name, res in (('tenryuu', (320, 240)), ('man', (200, 150)), ('clumsydragon', (256, 192))):
                osc, ob = oracle_bvh(name)
                sc = capi.DeviceScene(ctx, packed_scene(name))
                gb = capi.DeviceBVH(ctx, sc).build()
                cam = osc.demo_camera(*res)
                want, _ = ob.trace_primary(cam)
                for _ in range(2):
                    got = gb.generate_contacts_primary(as_capi_camera(cam))
                    assert got.tobytes() == want.tobytes(), (name, packet, budget, batch)
                assert ctx.last_trace_counters()['deferred_rays'] > 200
                if packet == '0':
                    rays = orc.generate_rays(cam)[::3].copy()                      # a ragged ray buffer through the same kernels
                    want_r, _ = ob.trace_rays(rays)
                    assert gb.generate_contacts_rays(rays).tobytes() == want_r.tobytes()
                gb.close(); sc.close()
        finally:
            ctx.close()


def test_search_tree_is_an_implementation_detail(monkeypatch):
    """primary rays walk the search tree by default and the reference tree with RTMI_SEARCH_TREE=0: same bytes, and both
    are the oracle's (which walks the reference tree in the reference's order)"""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    frames = {}
    for knob in ('1', '0'):
        monkeypatch.setenv('RTMI_SEARCH_TREE', knob)
        ctx = capi.Context(0)
        try:
            for name, res in (('tenryuu', (400, 300)), ('man', (320, 240)), ('cube', (64, 48))):
                osc, ob = oracle_bvh(name)
                sc = capi.DeviceScene(ctx, packed_scene(name))
                gb = capi.DeviceBVH(ctx, sc).build()
                cam = osc.demo_camera(*res)
                frames[(name, knob)] = gb.generate_contacts_primary(as_capi_camera(cam))
                st = gb.search_tree()
                if knob == '0':
                    assert st['depth'] == 0 and st['used_by_last_camera'] == 0
                elif name != 'cube':                          # 12 triangles: the reference tree is as shallow as it gets
                    assert 0 < st['depth'] < gb.info()['max_depth'] and st['used_by_last_camera'] == 1
                if knob == '1':
                    want, _ = ob.trace_primary(cam)
                    assert frames[(name, knob)].tobytes() == want.tobytes()
                gb.close(); sc.close()
        finally:
            ctx.close()
    for name in ('tenryuu', 'man', 'cube'):
        assert frames[(name, '1')].tobytes() == frames[(name, '0')].tobytes()


@pytest.mark.parametrize('name', ['man', 'clumsydragon'])
def test_camera_on_a_box_plane_falls_back_to_the_reference_tree(gpu_ctx, name):
    """camera x == 0 == a plane of leaf boxes of these models, and the middle pixel column has direction.x == 0: the slab
    test multiplies 0 by inf there (NaN), the one case the search-tree argument excludes.  The device notices the plane
    (used_by_last_camera == 2); on a y or z plane exactly those rays start at the reference tree's root, on an x plane -- this
    test -- they stay on the search tree since the end of round 3 in every kernel (the NaN on the first axis is absorbed:
    DESIGN.md section 2, child_valid_signed / child_mask_octant); contacts are the oracle's either way."""
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(160, 120).copy()
    planes = np.concatenate([ob.nodes['lo'][:ob.n, 0], ob.nodes['hi'][:ob.n, 0]])
    assert (planes == 0.0).any()
    cam['m'][0][3] = 0.0                                      # camera position x (Transform.h:282-295: translation column)
    import oracle as orc
    rays = orc.generate_rays(cam)
    assert (rays['direction'][:, 0] == 0.0).sum() == 120      # one pixel column
    want, _ = ob.trace_primary(cam)
    got = gb.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()
    assert gb.search_tree()['used_by_last_camera'] == 2       # plane noticed: that pixel column starts at the reference root
    cam2 = osc.demo_camera(160, 120)
    cam2['m'][0][3] = 0.123                                   # no plane there
    assert not (planes == np.float32(0.123)).any()
    want2, _ = ob.trace_primary(cam2)
    assert gb.generate_contacts_primary(as_capi_camera(cam2)).tobytes() == want2.tobytes()
    assert gb.search_tree()['used_by_last_camera'] == 1


@pytest.mark.parametrize('packet, budget', [('4', '1048576'), ('0', '1048576'), ('0', '64'), ('1', '256')])
@pytest.mark.parametrize('name', ['man', 'clumsydragon'])
def test_camera_on_a_box_plane_with_every_kernel(name, packet, budget, monkeypatch):
    """the same camera (x on a plane of leaf boxes, a pixel column with direction.x == 0) under every forced strategy: the
    four-wide records skip the boxes of inner children, which is sound on the search tree only; on an x plane every kernel keeps
    such rays on the search tree (sign-selected planes, NaN ignored), on a y or z plane they walk the reference tree with binary
    steps (the four-wide packet kernel hands them to the wave-per-ray kernel: test_cameras_on_planes_of_every_axis)."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_PACKET', packet)
    monkeypatch.setenv('RTMI_DEFER_BUDGET', budget)
    osc, ob = oracle_bvh(name)
    cam = osc.demo_camera(160, 120).copy()
    cam['m'][0][3] = 0.0
    want, _ = ob.trace_primary(cam)
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        for _ in range(2):
            got = gb.generate_contacts_primary(as_capi_camera(cam))
            assert got.tobytes() == want.tobytes()
        assert gb.search_tree()['used_by_last_camera'] == 2
        if packet == '4':
            # round 3: a 0 * inf on the X axis is absorbed by the staged test (tests/test_cpu_plane_rays.py),
            # so on its per-octant records the four-wide kernel keeps the pixel column with direction.x == 0 (round 2: 120 rays
            # handed to the wave-per-ray kernel, which walked the reference tree's depth-237 chains for them)
            assert ctx.last_trace_counters()['deferred_rays'] == 0
            # frame graphs: a captured frame leaves the wave-per-ray launch out where no ray needs it -- both cameras here; both
            # replay the oracle's contacts, also alternately on the same BVH (each replay restores its camera's records first)
            npx = 160 * 120
            cam2 = osc.demo_camera(160, 120).copy()
            cam2['m'][0][3] = 0.123
            want2, _ = ob.trace_primary(cam2)
            graphs = []
            for c in (cam, cam2):
                dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
                graphs.append((capi.FrameGraph(gb, sc, as_capi_camera(c), dc, dr), dc, dr))
            for _ in range(2):
                for (g, dc, dr), w in zip(graphs, (want, want2)):
                    ctx.fill_u32(dc, 0xDEADBEEF, npx * 8)
                    g.launch()
                    assert ctx.download(dc, npx, capi.CONTACT).tobytes() == w.tobytes()
            for g, dc, dr in graphs:
                g.close(); ctx.free(dc); ctx.free(dr)
        gb.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name,res', [('gardenlight', (160, 120)), ('tenryuu', (320, 240))])
def test_secondary_rays_two_bounces(gpu_ctx, name, res):
    """rt_generate_secondary_rays (extension, reference TODO 2.2): shadow rays from the primary contacts and reflections of
    those, each traced through the ray-buffer entry point: rays and contacts equal the oracle's at every bounce."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    ccam = as_capi_camera(cam)
    c0 = gb.generate_contacts_primary(ccam)
    want0, _ = ob.trace_primary(cam)
    assert c0.tobytes() == want0.tobytes()
    light = (0.0, 10.0, -330.0)
    # bounce 1: shadow rays towards the scene's light, from the camera's rays
    want_r1 = orc.generate_secondary_rays(want0, 0, light, 1e-3, cam=cam)
    got_r1 = capi.generate_secondary_rays(gpu_ctx, c0, 0, light, 1e-3, cam=ccam)
    assert len(got_r1) == len(want_r1) == int((want0['normal_t'][:, 3] > 0).sum()) > 0
    assert got_r1.tobytes() == want_r1.tobytes()
    want_c1, _ = ob.trace_rays(want_r1)
    got_c1 = gb.generate_contacts_rays(got_r1)
    assert got_c1.tobytes() == want_c1.tobytes()
    # bounce 2: reflections of the primary rays at the primary contacts, generated from an explicit ray buffer
    rays0 = orc.generate_rays(cam)
    want_r2 = orc.generate_secondary_rays(want0, 1, eps=1e-3, rays_in=rays0)
    got_r2 = capi.generate_secondary_rays(gpu_ctx, c0, 1, eps=1e-3, rays_in=rays0.view(capi.RAY))
    assert got_r2.tobytes() == want_r2.tobytes()
    want_c2, _ = ob.trace_rays(want_r2)
    assert gb.generate_contacts_rays(got_r2).tobytes() == want_c2.tobytes()
    # nothing hit -> nothing emitted
    none = np.zeros(7, dtype=capi.CONTACT)
    assert len(capi.generate_secondary_rays(gpu_ctx, none, 0, light, 1e-3, cam=ccam)) == 0


def _tmp_path(name):
    import tempfile
    return os.path.join(tempfile.mkdtemp(prefix='rtmi_bench_'), name)


def _bench_outputs(r, details):
    """bench.py prints ONE summary line (<= 4096 bytes: the driver keeps the tail of stdout) and writes the whole record to --details"""
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and len(lines[0]) <= 4096, (len(lines), [len(l) for l in lines])
    line = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline', 'verified'):
        assert k in line, k
    full = json.load(open(details))
    assert full['value'] == line['value'] and full['ms_per_step'] == line['ms_per_step']
    return line, full


def test_bench_group_rehearsal_assembles_identical_frames():
    """bench.py's N > 1 path on this one GPU: three ranks inside one process (RTMI_BENCH_GROUP=3; the library gathers by device
    copies because RCCL refuses several ranks on one device) trace and shade their tiles of all five 1080p frames through
    rt_multi_frame_*; rank 0 checks every assembled frame against the untiled one and prints the contract's JSON line, with the
    config-5 leg tiled the same way."""
    import subprocess
    import sys as _sys
    env = dict(os.environ, RTMI_BENCH_GROUP='3')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    details = _tmp_path('bench_group3.json')
    r = subprocess.run([_sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--check-frames', '--details', details],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'frame check ok (after the warm-up): 5 assembled frames equal the untiled ones' in r.stderr
    assert 'frame check ok (after the timed steps): 5 assembled frames equal the untiled ones' in r.stderr
    line, out = _bench_outputs(r, details)
    assert line['per_rank']['rows'] and len(line['per_rank']['rows']) == 3
    assert out['scaling'] == 'strong' and out['unit'] == 'Mrays/s' and out['value'] > 0
    assert 'REHEARSAL' in out['config']['parallelism']
    assert out['cpu_baseline'] is None and out['roofline']['traffic'] is None      # rank-0-at-N=1-only fields
    assert out['config5']['mrays_per_s'] > 0 and 'tiled over 3 rank' in out['config5']['workload']


def test_bench_tiled_path_with_one_rank():
    """The N > 1 code path of a job with one process per GPU, with a job of ONE rank (RTMI_BENCH_FORCE_TILED=1): communicator,
    frame graphs per send buffer, gather step, untile, compared with the untiled frames."""
    import subprocess
    import sys as _sys
    env = dict(os.environ, RTMI_BENCH_FORCE_TILED='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([_sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '7', '--warmup', '2', '--check-frames',
                        '--no-cpu-baseline', '--no-extra-legs'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'frame check ok (after the warm-up): 5 assembled frames equal the untiled ones' in r.stderr
    assert 'frame check ok (after the timed steps): 5 assembled frames equal the untiled ones' in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1 and json.loads(line[0])['value'] > 0


def test_bench_tiled_path_under_the_runtime_torch_bundles():
    """N > 1 loads torch first (torch.distributed), so the library, RCCL and torch share the HIP runtime torch bundles -- an older
    ROCm than the one librtmi.so is linked against.  One rank rehearses that combination: frame graphs are captured, their nodes
    noted (or not: then moved cameras fall back to the eager sequence), frames are assembled and checked."""
    import subprocess
    import sys as _sys
    env = dict(os.environ, RTMI_BENCH_FORCE_TILED='1', RTMI_BENCH_TORCH_FIRST='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RTMI_BENCH_GROUP'):
        env.pop(k, None)
    details = _tmp_path('bench_torch_first.json')
    r = subprocess.run([_sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '7', '--warmup', '2', '--check-frames',
                        '--no-cpu-baseline', '--no-extra-legs', '--details', details], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'frame check ok (after the timed steps): 5 assembled frames equal the untiled ones' in r.stderr
    _, line = _bench_outputs(r, details)
    assert line['value'] > 0 and line['per_rank'] and line['per_rank'][0]['trace_shade_ms_per_step'] > 0


_TORCH_FIRST_MULTI = r"""
import sys, numpy as np
import torch                                  # FIRST: the process runs on the HIP runtime torch bundles
torch.cuda.set_device(0)
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/oracle')
import bench
from openclraytracer_amd import capi
buf = bench.load_scene_bytes('tenryuu')
cam0 = bench.demo_camera(buf, 640, 360)
poses = bench.moving_cameras(buf, 640, 360, 8)
ctxs = [capi.Context(0) for _ in range(3)]
scenes = [capi.DeviceScene(c, buf) for c in ctxs]
bvhs = [capi.DeviceBVH(c, s).build() for c, s in zip(ctxs, scenes)]
comms = capi.Comm.create_group(ctxs)
frames = [capi.MultiFrame(c, b, s, cam0, 1, 1) for c, b, s in zip(comms, bvhs, scenes)]      # assembled on rank 1
ref_ctx = capi.Context(0); ref_scene = capi.DeviceScene(ref_ctx, buf); ref_bvh = capi.DeviceBVH(ref_ctx, ref_scene).build()
for k in [0, 3, 5, 3, 0, 7]:
    cam = poses[k:k + 1]
    for f in frames: f.set_camera(cam)
    for _ in range(3): capi.multi_frame_render(frames)
    frames[1].wait()
    got = frames[1].download_frame()
    want = capi.shade_blinn_phong(ref_ctx, ref_scene, cam, ref_bvh.generate_contacts_primary(cam)).reshape(-1, 4)
    assert got.tobytes() == want.tobytes(), k
print('torch-first multi frame ok', torch.version.hip)
"""


def test_moving_cameras_under_the_runtime_torch_bundles():
    """with torch imported first the whole process runs on the HIP runtime torch bundles (an older ROCm than librtmi.so is
    built against), where an executable graph whose kernel-node arguments were re-set replayed three times slower and five streams of
    moved graphs stalled.  Frame graphs no longer re-set node arguments there (kernel-only sequences are issued as plain launches from
    refreshed argument copies; other sequences render a moved camera eagerly when the runtime is the older one): the bench's moving-camera
    leg runs in that combination, verified against the oracle and within reach of the standing camera, and rt_multi_frame_set_camera
    follows a camera over an in-process group, frame for frame equal to the untiled calls."""
    import subprocess
    import sys as _sys
    env = dict(os.environ, RTMI_BENCH_TORCH_FIRST='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RTMI_BENCH_GROUP', 'RTMI_BENCH_FORCE_TILED'):
        env.pop(k, None)
    details = _tmp_path('bench_moving.json')
    r = subprocess.run([_sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-pmc', '--details', details],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    _, line = _bench_outputs(r, details)
    mv = line['moving_camera']
    assert line['verified']['moving_camera_frames'] == 30
    assert mv['step_ms'] < 2.0 * mv['same_poses_still_step_ms'], mv            # (round 3 under this runtime: three times slower, then a stall)
    for name, ps in mv['per_scene'].items():
        assert ps['moving_graph_ms_per_frame'] < 2.0 * ps['same_poses_still_graph_ms_per_frame'] + 0.05, (name, ps)
    r = subprocess.run([_sys.executable, '-c', _TORCH_FIRST_MULTI % dict(root=ROOT)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'torch-first multi frame ok' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_default_line_has_measured_rooflines():
    """the N = 1 line: the counters are collected by the run itself (rocprofv3 child processes), every fraction is <= 1 and
    recomputes from the fields beside it"""
    import subprocess
    import sys as _sys
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'RTMI_BENCH_GROUP', 'RTMI_BENCH_FORCE_TILED'):
        env.pop(k, None)
    # (the bench reports its progress on stderr; kept in a file under gpurun_out/ so that a run that stalls shows where)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    err_path = os.path.join(ROOT, 'gpurun_out', 'test_bench_default_line.err')
    with open(err_path, 'w') as err_file:
        # (no --details: the default path, gpurun_out/bench_details.json, is what the driver's run leaves behind)
        r = subprocess.run([_sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '3'], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=err_file, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + open(err_path).read()[-2000:]
    line, out = _bench_outputs(r, os.path.join(ROOT, 'gpurun_out', 'bench_details.json'))
    assert line['details_file'] == 'gpurun_out/bench_details.json'
    assert line['roofline']['frac'] == out['roofline']['frac'] and line['roofline']['traffic'] == out['roofline']['traffic']
    assert line['cpu_baseline']['value'] == out['cpu_baseline']['value'] and line['verified']['frames'] == 5
    assert line['legs']['config4']['mrays'] > 0 and line['legs']['config5']['mrays'] > 0
    rf, rv = out['roofline'], out['roofline_valu_issue']
    assert out['n_gpus'] == 1 and out['config']['rays_per_step'] == 5 * 1920 * 1080
    assert rf['bound'] == 'hbm' and rf['traffic'] is not None and 0 < rf['frac'] <= 1
    assert abs(rf['achieved'] - rf['traffic'] / (rf['avg_launch_ms'] * 1e-3) / 1e9) <= 0.05 * rf['achieved'] + 0.2
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    assert rv['bound'] == 'valu_issue' and 0 < rv['frac'] <= 1
    c = rv['counters_per_dispatch']
    assert abs(rv['frac'] - c['SQ_INSTS_VALU'] * 2 / (1024 * c['GRBM_GUI_ACTIVE'] / 8)) < 2e-3
    for need in ('SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INST_CYCLES_SALU', 'SQ_WAIT_INST_ANY'):
        assert need in c
    assert out['cpu_baseline']['value'] > 0 and out['config5']['mrays_per_s'] > 0
    for n, ps in out['per_scene'].items():
        assert ps['rebuild_plus_trace_ms'] > 0 and ps['strategy_after_rebuilds'] == ps['strategy'], n
        assert 0 < ps['valu_issue_frac'] <= 1
        assert ps['first_frame_ms'] > 0 and 0 < ps['calibration_launches'] <= 14 and ps['calibration_gpu_ms'] > 0
    mv = out['moving_camera']
    assert mv['step_ms'] > 0 and mv['same_poses_still_step_ms'] > 0 and set(mv['per_scene']) == set(out['per_scene'])
    for n, ps in mv['per_scene'].items():
        assert ps['launches']['with_preparation'] + ps['launches']['eager'] == 32, n       # every frame paid for its camera


@pytest.mark.parametrize('knobs', [dict(), dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='1048576'), dict(RTMI_PACKET='1', RTMI_DEFER_BUDGET='32'),
                                   dict(RTMI_PACKET='0', RTMI_DEFER_BUDGET='6', RTMI_LONG_BATCH='4')],
                         ids=['calibrated', 'packet4', 'packet32', 'lane6-batch4'])
@pytest.mark.parametrize('name', ['tenryuu', 'man', 'clumsydragon'])
def test_rotated_cameras_inside_outside_and_on_planes(name, knobs, monkeypatch):
    """seeded random camera poses (rotation about two axes; position outside, inside and exactly on planes of leaf boxes),
    odd resolutions, every first-phase kernel: contacts are the oracle's."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    osc, ob = oracle_bvh(name)
    rng = np.random.default_rng(len(name) * 7 + len(knobs))
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        lo = ob.nodes['lo'][:ob.n, :3].min(0)
        hi = ob.nodes['hi'][:ob.n, :3].max(0)
        for pose in range(3):
            res = (int(rng.integers(40, 280)), int(rng.integers(30, 180)))
            cam = osc.demo_camera(*res).copy()
            m = cam['m'][0]
            a, b = rng.uniform(-0.8, 0.8, 2)
            ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
            rot = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
            for r in range(3):
                for c in range(3):
                    m[r * 4 + c] = np.float32(rot[r, c])
            p = lo + (hi - lo) * rng.uniform(-0.3, 1.3, 3)
            if pose == 2:
                p = ob.nodes['lo'][int(rng.integers(0, ob.n)), :3]        # every coordinate on a plane of a leaf box
            m[3], m[7], m[11] = [np.float32(x) for x in p]
            want, _ = ob.trace_primary(cam)
            got = gb.generate_contacts_primary(as_capi_camera(cam))
            assert got.tobytes() == want.tobytes(), (name, knobs, pose, res)
        gb.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('name', ['gardenlight', 'tenryuu'])
def test_frame_graphs_replay_the_eager_frame(gpu_ctx, name):
    """rt_frame_graph: trace + shade of a frame captured into a HIP graph.  Replays write the same contacts and colours,
    bit for bit, as the eager calls, for the untiled frame, the tiled 16-byte layout and the tiled packed planes."""
    from openclraytracer_amd import capi
    osc, _ = oracle_bvh(name)
    sc, gb = gpu_bvh(gpu_ctx, name)
    res = (333, 200)
    cam = as_capi_camera(osc.demo_camera(*res))
    npx = res[0] * res[1]
    want_c = gb.generate_contacts_primary(cam)
    want_rgba = capi.shade_blinn_phong(gpu_ctx, sc, cam, want_c)
    # untiled
    dc, dr = gpu_ctx.alloc(npx * 32), gpu_ctx.alloc(npx * 16)
    g = capi.FrameGraph(gb, sc, cam, dc, dr)
    for _ in range(3):
        gpu_ctx.fill_u32(dc, 0xDEADBEEF, npx * 8); gpu_ctx.fill_u32(dr, 0xDEADBEEF, npx * 4)
        g.launch()
        assert gpu_ctx.download(dc, npx, capi.CONTACT).tobytes() == want_c.tobytes()
        assert gpu_ctx.download(dr, npx * 4, np.float32).tobytes() == want_rgba.tobytes()
    g.close(); gpu_ctx.free(dc); gpu_ctx.free(dr)
    # tiled: rank 1 of RT_TILING(3, 1), 16-byte and packed layouts
    til = capi.tiling(3, 1)
    slots0 = capi.tiles_owned_pixels(cam, 0, til)
    slots = capi.tiles_owned_pixels(cam, 1, til)
    dc, dr, dp = gpu_ctx.alloc(slots * 32), gpu_ctx.alloc(slots * 16), gpu_ctx.alloc(slots0 * 12)
    gb.generate_contacts_primary_tiled_d(cam, 1, til, dc)
    capi.shade_blinn_phong_tiled_d(gpu_ctx, sc, cam, 1, til, dc, dr)
    capi.shade_blinn_phong_tiled_packed_d(gpu_ctx, sc, cam, 1, til, dc, dp, slots0)
    want_tc = gpu_ctx.download(dc, slots, capi.CONTACT).tobytes()
    want_tr = gpu_ctx.download(dr, slots * 4, np.float32).tobytes()
    want_tp = gpu_ctx.download(dp, slots0 * 3, np.float32).reshape(3, slots0)[:, :slots].tobytes()
    g1 = capi.FrameGraph(gb, sc, cam, dc, dr, capi.FRAME_TILED, 1, til)
    g2 = capi.FrameGraph(gb, sc, cam, dc, dp, capi.FRAME_TILED_PACKED, 1, til, slots0)
    for _ in range(2):
        gpu_ctx.fill_u32(dc, 0xDEADBEEF, slots * 8); gpu_ctx.fill_u32(dr, 0xDEADBEEF, slots * 4)
        g1.launch()
        assert gpu_ctx.download(dc, slots, capi.CONTACT).tobytes() == want_tc
        assert gpu_ctx.download(dr, slots * 4, np.float32).tobytes() == want_tr
        gpu_ctx.fill_u32(dc, 0xDEADBEEF, slots * 8); gpu_ctx.fill_u32(dp, 0xDEADBEEF, slots0 * 3)
        g2.launch()
        assert gpu_ctx.download(dc, slots, capi.CONTACT).tobytes() == want_tc
        assert gpu_ctx.download(dp, slots0 * 3, np.float32).reshape(3, slots0)[:, :slots].tobytes() == want_tp
    g1.close(); g2.close()
    gpu_ctx.free(dc); gpu_ctx.free(dr); gpu_ctx.free(dp)


def test_frame_graph_error_behaviour(gpu_ctx):
    """frame graphs refuse what cannot be captured or makes no sense, with a message, and leave the context usable"""
    from openclraytracer_amd import capi
    osc, _ = oracle_bvh('cube')
    sc, gb = gpu_bvh(gpu_ctx, 'cube')
    cam = as_capi_camera(osc.demo_camera(64, 48))
    dc, dr = gpu_ctx.alloc(64 * 48 * 32), gpu_ctx.alloc(64 * 48 * 16)
    with pytest.raises(capi.RtError, match='unknown frame layout'):
        capi.FrameGraph(gb, sc, cam, dc, dr, layout=7)
    with pytest.raises(capi.RtError, match='rank/nranks invalid'):
        capi.FrameGraph(gb, sc, cam, dc, dr, layout=capi.FRAME_TILED, rank=3, nranks=2)
    with pytest.raises(capi.RtError, match='slots_per_rank too small'):
        capi.FrameGraph(gb, sc, cam, dc, dr, layout=capi.FRAME_TILED_PACKED, rank=0, nranks=1, slots_per_rank=16)
    with pytest.raises(capi.RtError, match='null argument'):
        capi.FrameGraph(gb, sc, cam, 0, dr)
    gpu_ctx.set_profiling(True)
    with pytest.raises(capi.RtError, match='profiling'):
        capi.FrameGraph(gb, sc, cam, dc, dr)
    gpu_ctx.set_profiling(False)
    g = capi.FrameGraph(gb, sc, cam, dc, dr)                      # and now it works
    g.launch()
    want = gb.generate_contacts_primary(cam)
    assert gpu_ctx.download(dc, 64 * 48, capi.CONTACT).tobytes() == want.tobytes()
    g.close(); g.close()                                          # closing twice is harmless
    gpu_ctx.free(dc); gpu_ctx.free(dr)


# ---- round 2: LITERAL_ZERO on the device, frame-graph validity, sticky error word ------------------------------------------

@pytest.mark.parametrize('name', ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon'])
def test_literal_zero_mode_on_the_device(gpu_ctx, name):
    """SURVEY.md 8(c) LITERAL_ZERO: the reference's guard as written; node bytes, sorted pairs and contacts equal the oracle's
    LITERAL_ZERO mode (whose radix-tree stage equals the reference kernel's own output, tests/test_gpu_reference_cl.py)."""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc = orc.Scene(packed_scene(name))
    ob = orc.BVH(osc, 1)
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))
    gb = capi.DeviceBVH(gpu_ctx, sc, mode=1).build()
    try:
        assert gb.read_morton_pairs().tobytes() == ob.pairs.tobytes()
        nodes = gb.read_nodes()
        diff = np.nonzero((nodes.view(np.uint8).reshape(-1, 48) != ob.nodes.view(np.uint8).reshape(-1, 48)).any(axis=1))[0]
        assert diff.size == 0, 'first differing node %d: %s vs %s' % (diff[0], nodes[diff[0]], ob.nodes[diff[0]])
        assert gb.search_tree()['depth'] == 0
        for res in ((64, 48), (320, 240)):
            cam = osc.demo_camera(*res)
            want, st = ob.trace_primary(cam)
            got = gb.generate_contacts_primary(as_capi_camera(cam))
            assert got.tobytes() == want.tobytes()
            if res == (64, 48) and name != 'tenryuu':
                # hit counts of this mode (tests/test_cpu_oracle_and_host.py::test_literal_zero_hit_counts explains SURVEY's 1552)
                assert st.hits == {'cube': 2209, 'gardenlight': 624, 'man': 1513, 'clumsydragon': 257}[name]
        rays = orc.generate_rays(osc.demo_camera(96, 64))
        want, _ = ob.trace_rays(rays)
        assert gb.generate_contacts_rays(rays.view(capi.RAY)).tobytes() == want.tobytes()
    finally:
        gb.close(); sc.close()


def test_frame_graph_goes_stale_instead_of_replaying_freed_buffers(gpu_ctx):
    """a graph holds the addresses of the context's deferred-ray lists; a later, larger trace on the same
    context reallocates them.  Launching the graph must then fail loudly, and a fresh graph must work."""
    from openclraytracer_amd import capi
    import oracle as orc
    from conftest import packed_scene
    ctx = capi.Context(0)                          # own context: its scratch starts small
    try:
        osc = orc.Scene(packed_scene('tenryuu'))
        ob = orc.BVH(osc, 0)
        sc = capi.DeviceScene(ctx, packed_scene('tenryuu'))
        gb = capi.DeviceBVH(ctx, sc).build()
        cam = osc.demo_camera(320, 240)
        ccam = as_capi_camera(cam)
        slots = capi.tiles_owned_pixels(ccam, 1, 4)
        d_c, d_o = ctx.alloc(slots * 32), ctx.alloc(slots * 16)
        g = capi.FrameGraph(gb, sc, ccam, d_c, d_o, capi.FRAME_TILED, 1, 4, slots)
        g.launch(); ctx.synchronize()
        big = osc.demo_camera(1920, 1080)
        gb.generate_contacts_primary(as_capi_camera(big))        # untiled full-HD frame on the same context: larger lists
        with pytest.raises(capi.RtError, match='stale'):
            g.launch()
        g.close()
        g2 = capi.FrameGraph(gb, sc, ccam, d_c, d_o, capi.FRAME_TILED, 1, 4, slots)
        g2.launch(); ctx.synchronize()
        want, _ = ob.trace_primary(cam)
        got = ctx.download(d_c, slots, capi.CONTACT)
        live = got['normalAndDistance'][:, 3] > 0
        assert live.any() and set(got['pixelIndex'][live]) <= set(np.nonzero(want['normal_t'][:, 3] > 0)[0])
        # a rebuild makes it stale as well
        gb.build()
        with pytest.raises(capi.RtError, match='stale'):
            g2.launch()
        g2.close(); ctx.free(d_c); ctx.free(d_o); gb.close(); sc.close()
    finally:
        ctx.close()


def test_frame_graph_survives_another_camera_on_the_same_bvh(gpu_ctx):
    """the per-camera marks live in the shared BVH records; an eager trace with another camera position
    rewrites them.  Replaying the graph must restore its own marks and give its own camera's contacts."""
    from openclraytracer_amd import capi
    import oracle as orc
    from conftest import packed_scene
    osc = orc.Scene(packed_scene('gardenlight'))
    ob = orc.BVH(osc, 0)
    sc = capi.DeviceScene(gpu_ctx, packed_scene('gardenlight'))
    gb = capi.DeviceBVH(gpu_ctx, sc).build()
    cam_a = osc.demo_camera(320, 240)
    cam_b = cam_a.copy()
    cam_b['m'][0][3] += 0.37; cam_b['m'][0][7] -= 1.25; cam_b['m'][0][11] += 4.0       # inside the model's box
    n = 320 * 240
    d_c, d_o = gpu_ctx.alloc(n * 32), gpu_ctx.alloc(n * 16)
    g = capi.FrameGraph(gb, sc, as_capi_camera(cam_a), d_c, d_o)
    want_b, _ = ob.trace_primary(cam_b)
    assert gb.generate_contacts_primary(as_capi_camera(cam_b)).tobytes() == want_b.tobytes()
    g.launch(); gpu_ctx.synchronize()
    want_a, _ = ob.trace_primary(cam_a)
    assert gpu_ctx.download(d_c, n, capi.CONTACT).tobytes() == want_a.tobytes()
    g.close(); gpu_ctx.free(d_c); gpu_ctx.free(d_o); gb.close(); sc.close()


# ---- round 2: the native multi-GPU frame driver (rt_comm_* / rt_multi_frame_*) ----------------------------------------------

@pytest.mark.parametrize('nranks,root_extra,root', [(1, 0, 0), (2, 1, 0), (3, 0, 0), (4, 1, 0), (6, 2, 0), (2, 1, 1), (4, 1, 3), (6, 2, 2)])
def test_native_multi_frame_group_on_one_device(nranks, root_extra, root):
    """one process, N contexts, all on device 0 (the group then exchanges by device copies: RCCL refuses two ranks on a device):
    tiling, frame graphs per send buffer, double buffering, stream order and the untile are the code a multi-GPU node runs.
    The assembled frame must equal the untiled one bit for bit, frame after frame, also when the camera of the group changes --
    and whichever rank assembles it (round 3: rt_multi_frame_create_rooted, tiles dealt by virtual rank)."""
    from openclraytracer_amd import capi
    import oracle as orc
    from conftest import packed_scene
    buf = packed_scene('tenryuu')
    osc = orc.Scene(buf)
    ctxs = [capi.Context(0) for _ in range(nranks)]
    try:
        scenes = [capi.DeviceScene(c, buf) for c in ctxs]
        bvhs = [capi.DeviceBVH(c, s).build() for c, s in zip(ctxs, scenes)]
        comms = capi.Comm.create_group(ctxs)
        assert [c.rank() for c in comms] == list(range(nranks)) and all(c.size() == nranks for c in comms)
        assert not comms[0].uses_rccl()
        for res in ((640, 360), (333, 251)):
            cam = as_capi_camera(osc.demo_camera(*res))
            frames = [capi.MultiFrame(comms[r], bvhs[r], scenes[r], cam, root_extra, root) for r in range(nranks)]
            lay = frames[0].layout()
            if res == (640, 360):
                for f in frames:
                    f.set_timing(True)
            assert lay['tiling'] == capi.tiling(nranks, root_extra if nranks > 1 else 0)
            assert sum(f.layout()['own_slots'] for f in frames) >= res[0] * res[1]
            # reference frame: untiled trace + shade on a context of its own
            rctx = capi.Context(0)
            rsc = capi.DeviceScene(rctx, buf)
            rb = capi.DeviceBVH(rctx, rsc).build()
            want = capi.shade_blinn_phong(rctx, rsc, cam, rb.generate_contacts_primary(cam))
            rb.close(); rsc.close(); rctx.close()
            for k in range(5):                                   # both send buffers, gathers overlapping the next trace
                capi.multi_frame_render(frames)
                if k in (0, 3, 4):
                    for f in frames:
                        f.wait()
                    got = frames[root].download_frame()
                    assert got.tobytes() == want.reshape(-1, 4).tobytes(), 'frame %d' % k
            assert frames[root].pixels_ptr() != 0 and all(f.pixels_ptr() == 0 for r, f in enumerate(frames) if r != root)
            if res == (333, 251):
                # the group's camera moves (rt_multi_frame_set_camera on every rank): the assembled frames follow, both send buffers
                for step in range(3):
                    cam2 = cam.copy()
                    cam2['viewTransform'][0][3] += np.float32(0.11 * (step + 1)); cam2['viewTransform'][0][11] += np.float32(0.07 * (step + 1))
                    rctx = capi.Context(0)
                    rsc = capi.DeviceScene(rctx, buf)
                    rb = capi.DeviceBVH(rctx, rsc).build()
                    want2 = capi.shade_blinn_phong(rctx, rsc, cam2, rb.generate_contacts_primary(cam2))
                    rb.close(); rsc.close(); rctx.close()
                    for f in frames:
                        f.set_camera(cam2)
                    capi.multi_frame_render(frames)
                    for f in frames:
                        f.wait()
                    assert frames[root].download_frame().tobytes() == want2.reshape(-1, 4).tobytes(), 'moved camera, frame %d' % step
            if res == (640, 360):
                for f in frames:
                    t = f.times()
                    assert t['frames'] == 5 and t['trace_ms'] > 0 and t['gather_ms'] >= 0
            assert comms[0].rccl_ranks() == 0                     # no RCCL communicator in a group that shares a device
            for f in frames:
                f.close()
        for c in comms:
            c.close()
        for b in bvhs:
            b.close()
        for s in scenes:
            s.close()
    finally:
        for c in ctxs:
            c.close()


def test_native_multi_frame_error_behaviour(gpu_ctx):
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import oracle as orc
    osc = orc.Scene(packed_scene('cube'))
    cam = as_capi_camera(osc.demo_camera(128, 96))
    sc, gb = gpu_bvh(gpu_ctx, 'cube')
    with pytest.raises(capi.RtError, match='rank/nranks invalid'):
        capi.Comm.create(gpu_ctx, None, 2, 2)
    with pytest.raises(capi.RtError, match='id is null'):
        capi.Comm.create(gpu_ctx, None, 0, 2)
    comm = capi.Comm.create(gpu_ctx, None, 0, 1)                 # a job of one rank needs no id and no RCCL
    other = capi.Context(0)
    osc2 = capi.DeviceScene(other, packed_scene('cube'))
    ob2 = capi.DeviceBVH(other, osc2).build()
    with pytest.raises(capi.RtError, match="communicator's context"):
        capi.MultiFrame(comm, ob2, osc2, cam)
    f = capi.MultiFrame(comm, gb, sc, cam)
    capi.multi_frame_render([f]); f.wait()
    want = capi.shade_blinn_phong(gpu_ctx, sc, cam, gb.generate_contacts_primary(cam))
    assert f.download_frame().tobytes() == want.reshape(-1, 4).tobytes()
    f.close(); comm.close(); ob2.close(); osc2.close(); other.close()


def test_cpp_demo_tiles_the_frame_over_several_ranks(gpu_ctx, tmp_path):
    """apps/rt_demo --gpus: the C++ MultiGPUFrame of the host mirror (three ranks sharing device 0 here) writes the same
    float4 frame as the single-device path, bit for bit"""
    import subprocess
    from openclraytracer_amd import capi
    osc, _ = oracle_bvh('gardenlight')
    sc, gb = gpu_bvh(gpu_ctx, 'gardenlight')
    demo = os.path.join(ROOT, 'apps', 'bin', 'rt_demo')
    raw = str(tmp_path / 'f.bin')
    r = subprocess.run([demo, '-winH', '240', '-winW', '320', '-accStruct', 'BVH', '-headersPath', 'unused', '-scene',
                        os.path.join('assets', 'scenes', 'gardenlight.txt'), '--frames', '3', '--devices', '0,0,0', '--out', raw],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'ranks 3' in r.stdout and 'device-copy gather' in r.stdout
    cam = as_capi_camera(osc.demo_camera(320, 240))
    want = capi.shade_blinn_phong(gpu_ctx, sc, cam, gb.generate_contacts_primary(cam))
    assert open(raw, 'rb').read() == want.reshape(-1, 4).tobytes()


# ---- round 2: refit by range queries (no serial climb) --------------------------------------------------------------------

@pytest.mark.parametrize('how', ['rq', 'climb', 'span'])
@pytest.mark.parametrize('name', ['gardenlight', 'tenryuu', 'man', 'clumsydragon'])
def test_range_query_refit_gives_the_same_node_bytes(name, how, monkeypatch):
    """RTMI_REFIT=rq: every inner box as the rightmost minimum / maximum over the node's span of sorted leaves (prefix / suffix
    folds per 64-leaf block + a sparse table) instead of climbing parent links; span (round 4, the default): the same fold read out of a
    segment tree over the sorted leaf boxes; climb: parent links.  Same 48-byte nodes, same depth, same contacts, whichever -- and the search
    tree, refitted the same way, nests tightly (test_search_tree_nests_tightly_over_the_reference_leaves runs on the default)."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_REFIT', how)
    osc, ob = oracle_bvh(name)
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        nodes = gb.read_nodes()
        diff = np.nonzero((nodes.view(np.uint8).reshape(-1, 48) != ob.nodes.view(np.uint8).reshape(-1, 48)).any(axis=1))[0]
        assert diff.size == 0, 'first differing node %d: %s vs %s' % (diff[0], nodes[diff[0]], ob.nodes[diff[0]])
        rc, depth = ob.check_hierarchy()
        assert rc == 0 and gb.info()['max_depth'] == depth
        cam = osc.demo_camera(320, 240)
        assert gb.generate_contacts_primary(as_capi_camera(cam)).tobytes() == ob.trace_primary(cam)[0].tobytes()
        gb.build()                                                  # and again on the kept buffers
        assert gb.read_nodes().tobytes() == ob.nodes.tobytes()
        gb.close(); sc.close()
    finally:
        ctx.close()


@pytest.mark.parametrize('how', ['rq', 'span'])
def test_range_query_refit_falls_back_on_nan_boxes(how, monkeypatch):
    """a NaN vertex makes min / max non-associative: the range-query / span refit detects it and the climbing refit runs instead (launched
    behind it, decided on the device)"""
    import oracle as orc
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_REFIT', how)
    buf = bytearray(packed_scene('gardenlight'))
    h = __import__('scene_pack').scene_header(bytes(buf))
    # first vertex of the first mesh: SceneHeader 96 + lights 16 each + spheres 16 + materials 96 each, ModelHeader 64, MeshHeader 48
    off = 96 + 16 * h['numberOfLights'] + 16 * h['numberOfSpheres'] + 96 * h['numberOfMaterials'] + 64 + 48
    buf[off + 4:off + 8] = np.float32(np.nan).tobytes()
    osc = orc.Scene(bytes(buf))
    ob = orc.BVH(osc, 0)
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, bytes(buf))
        gb = capi.DeviceBVH(ctx, sc).build()
        assert gb.read_nodes().tobytes() == ob.nodes.tobytes()
        gb.close(); sc.close()
    finally:
        ctx.close()


def test_32_bit_vertex_indices(gpu_ctx, tmp_path):
    """the reference's ushort indices (SceneBufferParser.h:160, Scene.cpp:430) wrap for a mesh with more than
    65 535 unique vertices.  With 32-bit indices (scene_pack_tool --index32 / Scene::setIndexWidth(32) / rt_scene_upload_indexed)
    one 90 000-vertex mesh traces exactly like the same triangles cut into two meshes that fit 16 bits; packed with 16-bit
    indices the big mesh is a different (wrapped) surface.  A small mesh gives identical BVH bytes in both widths."""
    import subprocess
    from openclraytracer_amd import capi
    import bench as bench_mod
    tool = os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')
    G = 300
    ys, xs = np.mgrid[0:G, 0:G]
    z = 0.6 * np.sin(xs * 0.07) * np.cos(ys * 0.05)
    verts = np.stack([xs * 0.1 - 15.0, ys * 0.1 - 15.0, z + 5.0], axis=-1).reshape(-1, 3)
    q = (ys[:-1, :-1] * G + xs[:-1, :-1]).reshape(-1)
    tris = np.concatenate([np.stack([q, q + 1, q + G], 1), np.stack([q + 1, q + G + 1, q + G], 1)], 0)
    order = np.argsort(tris[:, 0] // G, kind='stable')              # by row of the first vertex: both packings keep this triangle order
    tris = tris[order]

    def write_obj(path, groups):
        with open(path, 'w') as f:
            for v in verts:
                f.write('v %.6f %.6f %.6f\n' % tuple(v))
            for gname, tt in groups:
                f.write('g %s\n' % gname)
                for a, b, c in tt:
                    f.write('f %d %d %d\n' % (a + 1, b + 1, c + 1))
    half = len(tris) // 2
    half -= half % 2
    write_obj(tmp_path / 'one.obj', [('all', tris)])
    write_obj(tmp_path / 'two.obj', [('lower', tris[:half]), ('upper', tris[half:])])      # each part references < 65 536 vertices
    for n in ('one', 'two'):
        (tmp_path / (n + '.txt')).write_text('LIGHT 0.0 10.0 -330.0 1000.0\nMESH %s.obj\n' % n)
    pack = lambda n, *flags: (subprocess.check_call([tool, n + '.txt', n + ''.join(flags) + '.bin'] + list(flags), cwd=str(tmp_path)),
                              open(tmp_path / (n + ''.join(flags) + '.bin'), 'rb').read())[1]
    one32, one16, two16 = pack('one', '--index32'), pack('one'), pack('two')
    assert len(one32) - len(one16) == 2 * 3 * len(tris)

    def contacts(buf, index_bytes):
        sc = capi.DeviceScene(gpu_ctx, buf, index_bytes)
        gb = capi.DeviceBVH(gpu_ctx, sc).build()
        cam = bench_mod.demo_camera(buf, 640, 360)
        c = gb.generate_contacts_primary(cam)
        nodes = gb.read_nodes()
        gb.close(); sc.close()
        return c, nodes
    c32, _ = contacts(one32, 4)
    c2, _ = contacts(two16, 2)
    hit = c32['normalAndDistance'][:, 3] > 0
    assert hit.sum() > 50000
    assert c32.tobytes() == c2.tobytes()                               # same surface, same contacts
    c16, _ = contacts(one16, 2)
    assert c16.tobytes() != c32.tobytes()                              # ushort indices wrapped: another surface
    # a mesh that fits 16 bits: both widths give the same triangle table, hence the same BVH bytes and contacts
    small = tris[(tris < 60000).all(axis=1)][:20000]
    write_obj(tmp_path / 'small.obj', [('s', small)])
    (tmp_path / 'small.txt').write_text('LIGHT 0.0 10.0 -330.0 1000.0\nMESH small.obj\n')
    s32, s16 = pack('small', '--index32'), pack('small')
    (ca, na), (cb, nb) = contacts(s32, 4), contacts(s16, 2)
    assert ca.tobytes() == cb.tobytes() and na.tobytes() == nb.tobytes()
    with pytest.raises(capi.RtError, match='index width'):
        capi.DeviceScene(gpu_ctx, s16, 3)


# ---- round 3: cameras on planes of every axis, the camera that moves ---------------------------------------------------------

@pytest.mark.parametrize('knobs', [dict(), dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='1048576'), dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='64'),
                                   dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='1048576', RTMI_NO_OCTANT_RECORDS='1'),
                                   dict(RTMI_PACKET='0', RTMI_DEFER_BUDGET='32')],
                         ids=['calibrated', 'packet4', 'packet4-share64', 'packet4-general-records', 'lane32'])
@pytest.mark.parametrize('name', ['man', 'tenryuu'])
def test_cameras_on_planes_of_every_axis(name, knobs, monkeypatch):
    """cameras whose x, y or z coordinate equals a plane of leaf boxes AND that have rays with a zero direction component on that
    axis (tests/plane_rays.py: plane_camera), i.e. rays that multiply 0 by inf in the slab test.  On x the four-wide kernel keeps
    them on its per-octant records (NaN on the first axis is absorbed); on y and z, and without the per-octant records, they
    walk the reference tree.  The oracle walks the reference tree with the reference's staged test: contacts equal it bit for bit."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import plane_rays as pr
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    osc, ob = oracle_bvh(name)
    rng = np.random.default_rng(11)
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        for axis in (0, 1, 2):
            for trial in range(2):
                value = ob.nodes[('lo', 'hi')[trial]][int(rng.integers(0, ob.n)), axis]
                if value == 0:
                    value = ob.nodes['hi'][int(rng.integers(0, ob.n)), axis]
                cam = pr.plane_camera(osc, (96, 64), axis, value)
                rays = orc_rays(cam)
                with np.errstate(all='ignore'):
                    risky = int((~np.isfinite(np.float32(1.0) / rays['direction'][:, axis])).sum())
                assert risky >= 64
                want, _ = ob.trace_primary(cam)
                got = gb.generate_contacts_primary(as_capi_camera(cam))
                assert got.tobytes() == want.tobytes(), (name, knobs, axis, trial)
                if knobs.get('RTMI_PACKET') == '4':
                    handed_over = ctx.last_trace_counters()['deferred_rays']
                    general = 'RTMI_NO_OCTANT_RECORDS' in knobs
                    assert handed_over == (0 if (axis == 0 and not general) else risky), (axis, handed_over, risky)
        gb.close(); sc.close()
    finally:
        ctx.close()


def orc_rays(cam):
    import oracle as orc
    return orc.generate_rays(cam)


@pytest.mark.parametrize('knobs', [dict(), dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='128'), dict(RTMI_PACKET='0', RTMI_DEFER_BUDGET='24'),
                                   dict(RTMI_PACKET='1', RTMI_DEFER_BUDGET='256'), dict(RTMI_FUSED_FRAMES='0')],
                         ids=['calibrated', 'packet4-share128', 'lane24', 'packet256', 'unfused'])
@pytest.mark.parametrize('name,layout', [('tenryuu', 'untiled'), ('clumsydragon', 'packed'), ('gardenlight', 'tiled')])
def test_frame_graph_follows_a_moving_camera(name, layout, knobs, monkeypatch):
    """rt_frame_graph_set_camera (the Demo updates its camera before every RTFrame, OpenCLRayTracerDemo.cpp:275-276): one graph,
    a camera that translates and turns every frame -- outside and inside the model, on an x plane (stays in the graph), on a y
    plane with a pixel row of zero direction.y (served by the eager sequence), and back.  Every frame's contacts and colours
    equal the eager calls with that camera bit for bit (which the other tests pin on the oracle; every third frame is compared
    with the oracle directly), under every strategy, untiled and tiled; eager traces of other cameras on the same BVH in between
    do not disturb it."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import plane_rays as pr
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    osc, ob = oracle_bvh(name)
    res = (200, 120)
    npx = res[0] * res[1]
    rng = np.random.default_rng(3)
    lo = ob.nodes['lo'][:ob.n, :3].min(0)
    hi = ob.nodes['hi'][:ob.n, :3].max(0)

    def pose(i):
        cam = osc.demo_camera(*res).copy()
        m = cam['m'][0]
        if i == 4:                                        # x on a plane, a pixel column with direction.x == 0
            return pr.plane_camera(osc, res, 0, ob.nodes['lo'][17 % ob.n, 0])
        if i == 7:                                        # y on a plane, a pixel row with direction.y == 0
            return pr.plane_camera(osc, res, 1, ob.nodes['hi'][23 % ob.n, 1])
        a, b = rng.uniform(-0.9, 0.9, 2)
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        rot = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
        for r in range(3):
            for c in range(3):
                m[r * 4 + c] = np.float32(rot[r, c])
        p = lo + (hi - lo) * rng.uniform(-0.4, 1.4, 3)
        m[3], m[7], m[11] = [np.float32(x) for x in p]
        return cam

    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        cam0 = as_capi_camera(osc.demo_camera(*res))
        gb.generate_contacts_primary(cam0)               # the context's lists at full-frame size before a tiled graph holds their addresses
        if layout == 'untiled':
            rank, til, slots, kind = 0, 1, npx, capi.FRAME_UNTILED
            d_c, d_o = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
            out_words = npx * 4
        else:
            rank, til = 1, capi.tiling(3, 1)
            slots = capi.tiles_owned_pixels(cam0, rank, til)
            slots0 = capi.tiles_owned_pixels(cam0, 0, til)
            kind = capi.FRAME_TILED_PACKED if layout == 'packed' else capi.FRAME_TILED
            d_c = ctx.alloc(slots * 32)
            d_o = ctx.alloc(slots0 * 12 if layout == 'packed' else slots * 16)
            out_words = slots0 * 3 if layout == 'packed' else slots * 4
        g = capi.FrameGraph(gb, sc, cam0, d_c, d_o, kind, rank, til, slots0 if layout == 'packed' else 0)
        d_c2 = ctx.alloc(slots * 32)
        d_o2 = ctx.alloc(out_words * 4)
        kinds = set()
        for i in range(10):
            cam = pose(i)
            ccam = as_capi_camera(cam)
            # the eager frame of this camera into buffers of its own
            if layout == 'untiled':
                gb.generate_contacts_primary_d(ccam, d_c2)
                capi.shade_blinn_phong_frame_d(ctx, sc, ccam, d_c2, d_o2)
            else:
                gb.generate_contacts_primary_tiled_d(ccam, rank, til, d_c2)
                if layout == 'packed':
                    capi.shade_blinn_phong_tiled_packed_d(ctx, sc, ccam, rank, til, d_c2, d_o2, slots0)
                else:
                    capi.shade_blinn_phong_tiled_d(ctx, sc, ccam, rank, til, d_c2, d_o2)
            want_c = ctx.download(d_c2, slots, capi.CONTACT)
            want_o = ctx.download(d_o2, out_words, np.float32)
            if i % 3 == 0 and layout == 'untiled':
                assert want_c.tobytes() == ob.trace_primary(cam)[0].tobytes()
            if i % 2:                                     # somebody else's camera on the same BVH in between
                gb.generate_contacts_primary(as_capi_camera(pose(100 + i)))
            ctx.fill_u32(d_c, 0xDEADBEEF, slots * 8); ctx.fill_u32(d_o, 0xDEADBEEF, out_words)
            g.set_camera(ccam)
            g.launch()
            got_c = ctx.download(d_c, slots, capi.CONTACT)
            got_o = ctx.download(d_o, out_words, np.float32)
            kinds.add(g.last_launch())
            assert got_c.tobytes() == want_c.tobytes(), (name, layout, knobs, i)
            if layout == 'packed':
                assert got_o.reshape(3, slots0)[:, :slots].tobytes() == want_o.reshape(3, slots0)[:, :slots].tobytes(), (name, layout, knobs, i)
            else:
                assert got_o.tobytes() == want_o.tobytes(), (name, layout, knobs, i)
            # the camera stands still: the steady-state sequence, same bytes
            ctx.fill_u32(d_c, 0xDEADBEEF, slots * 8)
            g.launch()
            assert ctx.download(d_c, slots, capi.CONTACT).tobytes() == want_c.tobytes()
            kinds.add(g.last_launch())
        assert 1 in kinds and (0 in kinds or 2 in kinds)
        if knobs.get('RTMI_PACKET') == '4':
            assert kinds == {0, 1, 2}                    # the y-plane pose could not be served by the captured sequence
        with pytest.raises(capi.RtError, match='resolution'):
            g.set_camera(as_capi_camera(osc.demo_camera(64, 48)))
        g.close()
        for d in (d_c, d_o, d_c2, d_o2):
            ctx.free(d)
        gb.close(); sc.close()
    finally:
        ctx.close()


def test_cpp_reference_frame_loop(gpu_ctx, tmp_path):
    """apps/rt_demo --reference-frame 1: RTFrame as the reference wrote it (OpenCLRayTracerDemo.cpp:287-316) -- every contact of the
    primary buffer to the host, colours cleared, cpuShadeBlinnPhong per contact on the host (the mirror's CLData/Shading.h) -- gives
    the frame the device-shaded path gives, within 1e-5 relative (device powf), same lit pixels."""
    import subprocess
    demo = os.path.join(ROOT, 'apps', 'bin', 'rt_demo')
    frames = {}
    for mode in ('0', '1'):
        pfm = str(tmp_path / ('f%s.pfm' % mode))
        r = subprocess.run([demo, '-winH', '300', '-winW', '400', '-accStruct', 'BVH', '-headersPath', 'unused', '-scene',
                            os.path.join('assets', 'scenes', 'tenryuu.txt'), '--reference-frame', mode, '--out', pfm],
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ('host shading loop' in r.stdout) == (mode == '1')
        blob = open(pfm, 'rb').read()
        head = b'PF\n400 300\n-1.0\n'
        frames[mode] = np.frombuffer(blob, dtype='<f4', offset=len(head)).reshape(300, 400, 3)
    a, b = frames['0'], frames['1']
    assert ((a != 0).any(axis=2) == (b != 0).any(axis=2)).all() and (b != 0).any()
    err = np.abs(a - b) / np.maximum(np.abs(a), 1e-30)
    assert float(np.where(a == b, 0.0, err).max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('group', ['1', '16', '64'])
def test_block_start_order_changes_no_bit(group, monkeypatch):
    """The four-wide kernel starts a frame's blocks in the order of their step counts in earlier frames (rt_ctx_set_frames_in_flight,
    k_block_order): the order is a permutation of the blocks in groups of `group` neighbours, most expensive group first, and
    contacts and colours are the bits of the index-order frames -- eager frames over two refreshes of the order with a camera
    that moves, a frame graph replayed over two refreshes, and the sub-tree sharing (budget 128) included."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    monkeypatch.setenv('RTMI_PACKET', '4')
    monkeypatch.setenv('RTMI_DEFER_BUDGET', '128')
    monkeypatch.setenv('RTMI_BLOCK_ORDER_GROUP', group)
    monkeypatch.setenv('RTMI_BLOCK_ORDER_IN_FLIGHT', '0')     # the second context below (frames in flight) walks in index order
    name, res = 'man', (448, 320)
    npx = res[0] * res[1]
    osc, ob = oracle_bvh(name)
    cams = []
    for i in range(4):
        cam = osc.demo_camera(*res).copy()
        cam['m'][0][3] += np.float32(0.03 * i)
        cams.append(as_capi_camera(cam))
    ctxs = [capi.Context(0), capi.Context(0)]
    try:
        ctxs[1].set_frames_in_flight(2)
        scs = [capi.DeviceScene(c, packed_scene(name)) for c in ctxs]
        bvhs = [capi.DeviceBVH(c, s).build() for c, s in zip(ctxs, scs)]
        assert bvhs[1].block_order() is None
        g = int(group)
        for k in range(70):                                   # refreshes behind frames 1, 33 and 65
            cam = cams[(k // 9) % 4]
            a = bvhs[0].generate_contacts_primary(cam)
            if k in (0, 1, 2, 32, 33, 34, 66, 69):
                b = bvhs[1].generate_contacts_primary(cam)
                assert a.tobytes() == b.tobytes(), k
            if k in (1, 40, 69):
                order, cost = bvhs[0].block_order()
                blocks = len(order)
                assert blocks % 64 == 0 and blocks * 64 >= npx
                assert np.array_equal(np.sort(order), np.arange(blocks, dtype=np.uint32))
                assert np.array_equal(order.reshape(-1, g) - order.reshape(-1, g)[:, :1], np.tile(np.arange(g, dtype=np.uint32), (blocks // g, 1)))
                assert not np.array_equal(order, np.arange(blocks, dtype=np.uint32))
                assert cost.max() > 4 * max(1, int(np.median(cost)))          # what the order is for: a few blocks cost a multiple
        # the order in use is the one sorted from the costs counted 6 frames ago, same camera since frame 63: most expensive group first
        order, cost = bvhs[0].block_order()
        sums = cost.astype(np.int64).reshape(-1, g).sum(1)
        first = order[::g] // g
        assert sums[first[0]] == sums.max()
        assert sums[first[:len(first) // 8]].mean() > sums.mean()
        # a frame graph (its own order, sorted behind its first replay and every 32 after that) against the index-order graph
        outs = []
        for c, s, b in zip(ctxs, scs, bvhs):
            d_c, d_o = c.alloc(npx * 32), c.alloc(npx * 16)
            fg = capi.FrameGraph(b, s, cams[0], d_c, d_o)
            frames = []
            for k in range(36):
                fg.launch()
                if k in (0, 1, 35):
                    c.synchronize()
                    frames.append((c.download(d_c, npx * 32, np.uint8).tobytes(), c.download(d_o, npx * 16, np.uint8).tobytes()))
            outs.append(frames)
            fg.close()
            c.free(d_c); c.free(d_o)
        assert outs[0] == outs[1]
        assert outs[0][0] == outs[0][2]
        assert outs[0][0][0] == bvhs[1].generate_contacts_primary(cams[0]).tobytes()
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['man', 'tenryuu', 'clumsydragon', 'cube'])
def test_search_tree_nests_tightly_over_the_reference_leaves(name, monkeypatch):
    """What DESIGN.md section 2 asks of the search tree and nothing more: the SAME n leaf boxes as the reference tree (bit for bit),
    every leaf below the root exactly once, every inner box the plain union of its children's boxes (rt_bvh_read_search_nodes) -- in
    particular no maximum pulled up to FLT_MIN as in the reference's own refit (BVH.h:309-310), which the reference tree keeps --
    and, with RTMI_SEARCH_TREE_BOXES=reference, round 2's clamped boxes, the same contacts either way."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc, ob = oracle_bvh(name)
    n = ob.n
    cam = as_capi_camera(osc.demo_camera(320, 200))
    frames = {}
    for boxes in ('tight', 'reference'):
        monkeypatch.setenv('RTMI_SEARCH_TREE_BOXES', boxes)
        ctx = capi.Context(0)
        try:
            sc = capi.DeviceScene(ctx, packed_scene(name))
            gb = capi.DeviceBVH(ctx, sc).build()
            ref, sn = gb.read_nodes(), gb.read_search_nodes()
            for f in ('boundsMin', 'boundsMax'):                                            # the reference's leaves (their parents differ, of course)
                assert ref[f][:n].tobytes() == sn[f][:n].tobytes()
            assert np.array_equal(ref['data'][:n, 1:], sn['data'][:n, 1:])
            assert ref.tobytes() == ob.nodes.tobytes()                                    # ... and the reference tree is still the oracle's
            lo, hi = sn['boundsMin'][:, :3].copy(), sn['boundsMax'][:, :3].copy()
            a, b = sn['data'][n:, 1].astype(np.int64), sn['data'][n:, 2].astype(np.int64)
            seen = np.bincount(np.concatenate([a, b]), minlength=2 * n - 1)
            assert (seen[:n] == 1).all() and seen[n] == 0 and (seen[n + 1:] == 1).all()   # a tree: every node but the root has one parent
            want_lo, want_hi = np.minimum(lo[a], lo[b]), np.maximum(hi[a], hi[b])
            if boxes == 'tight':
                assert np.array_equal(lo[n:], want_lo) and np.array_equal(hi[n:], want_hi)
                if name != 'cube':
                    assert (hi[n:] < 0).any()                                             # boxes of the negative half end where their triangles do
            else:
                fmin = np.float32(np.finfo(np.float32).tiny)
                assert np.array_equal(lo[n:], want_lo) and np.array_equal(hi[n:], np.maximum(want_hi, fmin))
            frames[boxes] = gb.generate_contacts_primary(cam).tobytes()
            gb.close(); sc.close()
        finally:
            ctx.close()
    assert frames['tight'] == frames['reference'] == ob.trace_primary(osc.demo_camera(320, 200))[0].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize('knobs', [dict(), dict(RTMI_PACKET='4', RTMI_DEFER_BUDGET='32'), dict(RTMI_PACKET='0', RTMI_DEFER_BUDGET='16')],
                         ids=['calibrated', 'packet4-share32', 'lane16'])
def test_throughput_policy_changes_no_bit(knobs, monkeypatch):
    """rt_ctx_set_frames_in_flight(n > 1): workgroups of one wave for the first-phase kernels (their LDS stacks sized per launch), the
    sharing handicap in the calibration, a calibration of its own -- and the contacts and colours of the default policy, eager and
    through a frame graph (which, holding kernels only, is issued as plain launches), byte for byte."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    res = (416, 296)
    npx = res[0] * res[1]
    for name in ('tenryuu', 'clumsydragon'):
        osc, ob = oracle_bvh(name)
        cam = as_capi_camera(osc.demo_camera(*res))
        want = ob.trace_primary(osc.demo_camera(*res))[0].tobytes()
        frames = []
        for n_in_flight in (1, 3):
            ctx = capi.Context(0)
            try:
                ctx.set_frames_in_flight(n_in_flight)
                sc = capi.DeviceScene(ctx, packed_scene(name))
                gb = capi.DeviceBVH(ctx, sc).build()
                eager = gb.generate_contacts_primary(cam).tobytes()
                d_c, d_o = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
                fg = capi.FrameGraph(gb, sc, cam, d_c, d_o)
                for _ in range(3):
                    fg.launch()
                ctx.synchronize()
                frames.append((eager, ctx.download(d_c, npx * 32, np.uint8).tobytes(), ctx.download(d_o, npx * 16, np.uint8).tobytes()))
                fg.close(); ctx.free(d_c); ctx.free(d_o); gb.close(); sc.close()
            finally:
                ctx.close()
        assert frames[0] == frames[1]
        assert frames[0][0] == want and frames[0][1] == want


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['man', 'clumsydragon'])
def test_eager_frames_after_the_host_knows_the_planes(name, monkeypatch):
    """From the 33rd eager four-wide frame on a build the host looks the camera up in the planes of the leaf boxes itself
    (ensure_host_planes) and leaves the wave-per-ray launch and the per-camera marks out when the camera sits on no plane that
    matters.  Cameras off every plane, on an x plane (stays on the search tree), on a y and on a z plane (rays with a zero
    direction component there are handed over: the launch must be there), forty frames each and interleaved: every frame is the
    oracle's, and a rebuild starts the count again."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    import plane_rays as pr
    monkeypatch.setenv('RTMI_PACKET', '4')
    monkeypatch.setenv('RTMI_DEFER_BUDGET', '1048576')
    osc, ob = oracle_bvh(name)
    res = (160, 120)
    n = ob.n
    cams = [osc.demo_camera(*res)]
    cx = osc.demo_camera(*res).copy(); cx['m'][0][3] = 0.0; cams.append(cx)                       # x on a plane of these models
    cams.append(pr.plane_camera(osc, res, 1, ob.nodes['hi'][23 % n, 1]))                          # y on a plane, a pixel row with direction.y == 0
    cams.append(pr.plane_camera(osc, res, 2, ob.nodes['lo'][17 % n, 2]))                          # z on a plane, every ray with direction.z == 0
    want = [ob.trace_primary(c)[0].tobytes() for c in cams]
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        gb = capi.DeviceBVH(ctx, sc).build()
        for round_ in range(2):
            for k, c in enumerate(cams):
                for _ in range(40):
                    assert gb.generate_contacts_primary(as_capi_camera(c)).tobytes() == want[k], (round_, k)
            for i in range(16):                                                                    # interleaved
                k = (i * 7 + 3) % len(cams)
                assert gb.generate_contacts_primary(as_capi_camera(cams[k])).tobytes() == want[k], (round_, 'mixed', k)
            assert gb.host_plane_reads() == round_ + 1                                              # once per build, from its 33rd frame on
            gb.build()                                                                             # the count starts again
        # a scene that is rebuilt every frame never pays for the read-back
        reads = gb.host_plane_reads()
        for _ in range(12):
            gb.build()
            assert gb.generate_contacts_primary(as_capi_camera(cams[0])).tobytes() == want[0]
        assert gb.host_plane_reads() == reads
    finally:
        ctx.close()


@pytest.mark.gpu
def test_tile_tables_of_many_frame_shapes_in_one_context():
    """The first-phase kernels read where a rank's tile lies from a per-shape table (work_to_pixel_block / tile_table): a context keeps
    up to 16 of them, keyed by the whole (resolution, rank, tiling) tuple -- among the shapes here a pair that a hashed key of shifted
    fields would confuse (resY 120 with 0 extra tiles against resY 121 with 1) -- and computes the place per lane beyond that.  Every
    rank's compact contacts, put back in place, are the untiled frame's."""
    from openclraytracer_amd import capi
    from conftest import packed_scene
    osc, ob = oracle_bvh('gardenlight')
    shapes = [((200, 120), 3, 0), ((200, 121), 3, 1), ((200, 121), 3, 0), ((200, 120), 3, 1), ((333, 200), 2, 2), ((130, 70), 4, 0),
              ((130, 70), 4, 1), ((64, 64), 2, 0), ((65, 64), 2, 0), ((640, 360), 8, 1)]
    ctx = capi.Context(0)
    try:
        sc = capi.DeviceScene(ctx, packed_scene('gardenlight'))
        gb = capi.DeviceBVH(ctx, sc).build()
        for round_ in range(2):                                   # second round: every table exists (or never will)
            for res, n_ranks, extra in shapes:
                cam = as_capi_camera(osc.demo_camera(*res))
                til = capi.tiling(n_ranks, extra)
                full = gb.generate_contacts_primary(cam)
                if round_ == 0:
                    assert full.tobytes() == ob.trace_primary(osc.demo_camera(*res))[0].tobytes()
                rank_of, slot_of = capi.tile_slot_map(res[0], res[1], til)
                for r in range(n_ranks):
                    slots = capi.tiles_owned_pixels(cam, r, til)
                    d = ctx.alloc(max(slots, 1) * 32)
                    ctx.fill_u32(d, 0xDEADBEEF, max(slots, 1) * 8)
                    gb.generate_contacts_primary_tiled_d(cam, r, til, d)
                    compact = ctx.download(d, slots, capi.CONTACT)
                    ctx.free(d)
                    mine = rank_of == r
                    assert compact[slot_of[mine]].tobytes() == full[mine].tobytes(), (res, n_ranks, extra, r)
    finally:
        ctx.close()
