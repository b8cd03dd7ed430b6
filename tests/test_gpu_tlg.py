"""GPU parity of the Two-Level Grid (rt_tlg_*, through the C ABI) against the oracle: every build product and every
contact bit for bit (BASELINE.json configs[3] path: TLG build + trace)."""
import numpy as np
import pytest

from conftest import MODELS, as_capi_camera, packed_scene

pytestmark = pytest.mark.gpu

_otlg, _gtlg = {}, {}


def oracle_tlg(name, dens=(2.0, 2.0)):
    import oracle as orc
    key = (name, dens)
    if key not in _otlg:
        sc = orc.Scene(packed_scene(name))
        _otlg[key] = (sc, orc.TLG(sc, *dens))
    return _otlg[key]


def gpu_tlg(ctx, name, dens=(2.0, 2.0)):
    from openclraytracer_amd import capi
    key = (name, dens)
    if key not in _gtlg:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        _gtlg[key] = (sc, capi.DeviceTLG(ctx, sc, *dens).build())
    return _gtlg[key]


@pytest.mark.parametrize('name', MODELS)
def test_build_products_bit_exact(gpu_ctx, name):
    _, og = oracle_tlg(name)
    _, gg = gpu_tlg(gpu_ctx, name)
    info = gg.info()
    for k in info:
        assert info[k] == og.info[k], k
    T = type(gg)
    assert gg.read(T.GRID_DATA).tobytes() == og.grid.tobytes()
    assert gg.read(T.TOP_PAIRS).tobytes() == og.top_pairs.tobytes()
    assert gg.read(T.CELL_RANGES).tobytes() == og.cell_ranges.tobytes()
    assert gg.read(T.TOP_CELLS).tobytes() == og.top_cells.tobytes()
    assert gg.read(T.LEAF_PAIRS).tobytes() == og.leaf_pairs.tobytes()
    assert gg.read(T.LEAF_RANGES).tobytes() == og.leaf_ranges.tobytes()


@pytest.mark.parametrize('name,res', [('cube', (64, 48)), ('gardenlight', (64, 48)), ('gardenlight', (333, 251)),
                                      ('tenryuu', (640, 480)), ('man', (640, 480)), ('clumsydragon', (640, 480))])
def test_trace_primary_bit_exact(gpu_ctx, name, res):
    osc, og = oracle_tlg(name)
    _, gg = gpu_tlg(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    want, _ = og.trace_primary(cam)
    got = gg.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize('name', MODELS)
def test_trace_primary_full_hd_bit_exact(gpu_ctx, name):
    osc, og = oracle_tlg(name)
    _, gg = gpu_tlg(gpu_ctx, name)
    cam = osc.demo_camera(1920, 1080)
    want, _ = og.trace_primary(cam)
    got = gg.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()


def test_other_densities_and_rebuild(gpu_ctx):
    """setTopLevelDensity / setLeafDensity take effect at initializeFrame (TwoLevelGridManager.h:112-118)"""
    from openclraytracer_amd import capi
    name = 'gardenlight'
    osc, _ = oracle_tlg(name)
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))
    g = capi.DeviceTLG(gpu_ctx, sc)
    try:
        for dens in ((2.0, 2.0), (5.0, 1.2), (0.7, 3.5), (2.0, 2.0)):
            _, og = oracle_tlg(name, dens)
            gpu_ctx.check(gpu_ctx.lib.rt_tlg_set_densities(g.h, dens[0], dens[1]))
            g.build()
            assert g.info()['leaf_pairs_count'] == og.info['leaf_pairs_count']
            assert g.read(g.LEAF_PAIRS).tobytes() == og.leaf_pairs.tobytes()
            cam = osc.demo_camera(200, 150)
            want, _ = og.trace_primary(cam)
            assert g.generate_contacts_primary(as_capi_camera(cam)).tobytes() == want.tobytes()
    finally:
        g.close(); sc.close()


def test_ray_buffer_random_and_degenerate(gpu_ctx):
    """generateContacts(CLBuffer& rays, ...) (TwoLevelGridKernels.cl:251-270): incoherent rays, origins inside and outside
    the grid, axis-parallel / zero / NaN directions, a count that is not a multiple of 64"""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, og = oracle_tlg('tenryuu')
    _, gg = gpu_tlg(gpu_ctx, 'tenryuu')
    rng = np.random.default_rng(9)
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
    rays['direction'][5, :3] = (1e-42, 0.6, 0.8)         # denormal component: 1/d overflows
    rays['direction'][6, :3] = (-0.0, -0.6, 0.8)
    rays['origin'][7, :3] = (0, 0.5, 0)
    want, st = og.trace_rays(rays)
    got = gg.generate_contacts_rays(rays.view(capi.RAY))
    assert got.tobytes() == want.tobytes()
    assert st['hits'] > 100
    assert len(gg.generate_contacts_rays(rays[:0].view(capi.RAY))) == 0
    assert gg.generate_contacts_rays(rays[:1].view(capi.RAY)).tobytes() == want[:1].tobytes()


@pytest.mark.parametrize('nranks', [2, 3])
def test_tiled_frame_equals_untiled(gpu_ctx, nranks):
    from openclraytracer_amd import capi, frame_tiles
    osc, og = oracle_tlg('clumsydragon')
    _, gg = gpu_tlg(gpu_ctx, 'clumsydragon')
    cam = osc.demo_camera(500, 300)
    ccam = as_capi_camera(cam)
    want, _ = og.trace_primary(cam)
    npx = 500 * 300
    frame = np.zeros(npx, dtype=capi.CONTACT)
    owner, slot = frame_tiles.slot_map(500, 300, nranks)
    for rank in range(nranks):
        slots = capi.tiles_owned_pixels(ccam, rank, nranks)
        d = gpu_ctx.alloc(max(slots, 1) * 32)
        try:
            gpu_ctx.fill_u32(d, 0xDEADBEEF, max(slots, 1) * 8)
            gg.generate_contacts_primary_tiled_d(ccam, rank, nranks, d)
            compact = gpu_ctx.download(d, slots, capi.CONTACT)
        finally:
            gpu_ctx.free(d)
        mine = owner == rank
        frame[mine] = compact[slot[mine]]
    assert frame.tobytes() == want.tobytes()


def test_error_behaviour(gpu_ctx):
    from openclraytracer_amd import capi
    sc = capi.DeviceScene(gpu_ctx, packed_scene('cube'))
    g = capi.DeviceTLG(gpu_ctx, sc)
    try:
        with pytest.raises(capi.RtError, match='initializeFrame must precede construct'):
            g.construct()
        g.initialize_frame()
        cam = as_capi_camera(oracle_tlg('cube')[0].demo_camera(8, 8))
        with pytest.raises(capi.RtError, match='construct must precede generateContacts'):
            g.generate_contacts_primary(cam)
        with pytest.raises(capi.RtError, match='densities must be positive'):
            gpu_ctx.check(gpu_ctx.lib.rt_tlg_set_densities(g.h, 0.0, 2.0))
        g.construct()
        with pytest.raises(capi.RtError, match='size mismatch'):
            out = np.zeros(3, dtype=np.uint8)
            gpu_ctx.check(gpu_ctx.lib.rt_tlg_read(g.h, g.TOP_CELLS, out.ctypes.data, 3))
    finally:
        g.close(); sc.close()


@pytest.mark.parametrize('name', ['gardenlight', 'tenryuu', 'clumsydragon'])
def test_rebuild_runs_without_mid_build_readbacks_and_gives_the_same_products(gpu_ctx, name):
    """dynamic scenes rebuild per frame (OpenCLRayTracerDemo.cpp:230-234): the second construct() runs on the first one's array
    sizes with the counts kept on the device (no host round trip until the end) and must leave exactly the same products;
    when the sizes no longer fit (other densities) the sequence with read-backs takes over."""
    from openclraytracer_amd import capi
    import oracle as orc
    osc = orc.Scene(packed_scene(name))
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))
    gg = capi.DeviceTLG(gpu_ctx, sc).build()
    T = type(gg)

    def check(og):
        info = gg.info()
        for k in info:
            assert info[k] == og.info[k], k
        for what, want in ((T.TOP_PAIRS, og.top_pairs), (T.CELL_RANGES, og.cell_ranges), (T.TOP_CELLS, og.top_cells),
                           (T.LEAF_PAIRS, og.leaf_pairs), (T.LEAF_RANGES, og.leaf_ranges)):
            assert gg.read(what).tobytes() == want.tobytes(), what
        cam = osc.demo_camera(160, 120)
        assert gg.generate_contacts_primary(as_capi_camera(cam)).tobytes() == og.trace_primary(cam)[0].tobytes()

    og = orc.TLG(osc)
    check(og)
    assert gg.rebuilds_without_readback() == 0
    for k in range(2):
        gg.build()
        assert gg.rebuilds_without_readback() == k + 1
        check(og)
    gg.set_densities(3.1, 1.3)                       # other pair counts: the previous sizes do not fit
    gg.build()
    check(orc.TLG(osc, 3.1, 1.3))
    before = gg.rebuilds_without_readback()
    gg.build()
    assert gg.rebuilds_without_readback() == before + 1
    check(orc.TLG(osc, 3.1, 1.3))
    gg.close(); sc.close()


@pytest.mark.parametrize('name', ['gardenlight', 'tenryuu'])
def test_walk_records_are_in_use_and_change_no_contact(gpu_ctx, name):
    """Round 5: the grid walk divides by short multiply-add sequences on the walk's own cell / plane records (rt_tlg_walk_records,
    tlg_kernels.hip k_tlg_trace_records; tools/micro/div_check.hip compares the sequences with the IEEE division).  The bundled scenes
    qualify; with the records switched off (RTMI_TLG_WALK_RECORDS=0, a process of its own: the knob is read by rt_ctx_create) the
    reference's divisions run and every contact is the same -- and the oracle's."""
    import hashlib, os, subprocess, sys, tempfile
    from conftest import ROOT
    osc, og = oracle_tlg(name)
    _, gg = gpu_tlg(gpu_ctx, name)
    assert gg.walk_records()
    cam = osc.demo_camera(640, 480)
    want, _ = og.trace_primary(cam)
    assert gg.generate_contacts_primary(as_capi_camera(cam)).tobytes() == want.tobytes()
    with tempfile.TemporaryDirectory(prefix='rtmi_test_') as d:
        open(os.path.join(d, 's.bin'), 'wb').write(packed_scene(name))
        np.save(os.path.join(d, 'cam.npy'), cam.view(np.uint8))
        code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); from openclraytracer_amd import capi; ctx = capi.Context(0); "
                "sc = capi.DeviceScene(ctx, open(sys.argv[1], 'rb').read()); g = capi.DeviceTLG(ctx, sc).build(); "
                "c = g.generate_contacts_primary(np.load(sys.argv[2]).view(capi.CAMERA)); "
                "print('RESULT', int(g.walk_records()), hashlib.sha256(c.tobytes()).hexdigest())" % ROOT)
        # ... and without the camera-relative triangle records (RTMI_TLG_TRICAM=0): the records stay in use, every wave walks lane by lane on the
        # scene's triangle table (the packet walk of primary rays needs the camera's tables)
        for knob, want_flag in (('RTMI_TLG_WALK_RECORDS', '0'), ('RTMI_TLG_TRICAM', '1')):
            r = subprocess.run([sys.executable, '-c', code, os.path.join(d, 's.bin'), os.path.join(d, 'cam.npy')],
                               env=dict(os.environ, **{knob: '0'}), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            flag, digest = [l for l in r.stdout.splitlines() if l.startswith('RESULT')][0].split()[1:]
            assert flag == want_flag and digest == hashlib.sha256(want.tobytes()).hexdigest(), knob


def test_walk_records_rays_at_the_edges_of_what_qualifies(gpu_ctx):
    """Waves of rays that just qualify for the short division sequences, that just do not, and mixtures: direction components of exactly
    2^-20 and one ulp below, 2^20 and above (unnormalised directions), origins at 2^30 and beyond, origins exactly ON planes of the
    top-level grid (numerators of exactly zero), origins inside cells, and rays parallel to an axis in every other lane.  The oracle's
    contacts, bit for bit."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, og = oracle_tlg('tenryuu')
    _, gg = gpu_tlg(gpu_ctx, 'tenryuu')
    assert gg.walk_records()
    grid = og.grid
    lo = grid['lo'][0][:3].astype(np.float32)
    step = grid['step'][0].astype(np.float32)
    res = grid['res'][0]
    rng = np.random.default_rng(21)
    groups = []

    def group(n=64):
        r = np.zeros(n, dtype=orc.RAY)
        r['origin'][:, :3] = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        r['direction'][:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        return r
    tiny = np.float32(2.0 ** -20)
    for k in range(3):
        g = group(); g['direction'][:, k] = tiny * np.where(rng.integers(0, 2, 64) == 0, 1, -1).astype(np.float32); groups.append(g)      # qualifies
        g = group(); g['direction'][:, k] = np.nextafter(tiny, np.float32(0)); groups.append(g)                                          # does not
        g = group(); g['direction'][::2, k] = 0.0; groups.append(g)                                                                      # every other lane parallel to a plane
        g = group(); g['direction'][:, :3] *= np.float32(2.0 ** 20) / np.abs(g['direction'][:, :3]).max(axis=1, keepdims=True); groups.append(g)   # largest component exactly 2^20
        g = group(); g['direction'][:, :3] *= np.float32(2.0 ** 21); groups.append(g)
        # origins exactly on top-level planes of axis k (inside the grid and on its faces), directions arbitrary
        g = group(); g['origin'][:, k] = (lo[k] + rng.integers(0, res[k] + 1, 64).astype(np.float32) * step[k]).astype(np.float32); groups.append(g)
        # far origins looking at the model: 2^30 qualifies, 2^31 does not
        for far in (2.0 ** 30, 2.0 ** 31, 1.0e6):
            g = group(); g['origin'][:, :3] = 0; g['origin'][:, k] = np.float32(far)
            tgt = rng.uniform(-0.5, 0.5, size=(64, 3)).astype(np.float32)
            d = (tgt.astype(np.float64) - g['origin'][:, :3].astype(np.float64)); d /= np.linalg.norm(d, axis=1, keepdims=True)
            g['direction'][:, :3] = d.astype(np.float32); groups.append(g)
    g = group(640); groups.append(g)
    rays = np.concatenate(groups)
    rays['idx'] = np.arange(len(rays))
    want, st = og.trace_rays(rays)
    got = gg.generate_contacts_rays(rays.view(capi.RAY))
    assert got.tobytes() == want.tobytes()
    assert st['hits'] > 50


def test_walk_records_are_refused_by_grids_that_do_not_qualify(gpu_ctx):
    """More planes than the trace kernel's table holds (a top-level density that gives ~200 cells per axis): the build keeps no records,
    the walk is the reference's, the contacts are the oracle's."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc = orc.Scene(packed_scene('cube'))
    dens = (700000.0, 0.5)
    og = orc.TLG(osc, *dens)
    assert int(og.grid['res'][0].sum()) > 512
    sc = capi.DeviceScene(gpu_ctx, packed_scene('cube'))
    g = capi.DeviceTLG(gpu_ctx, sc, *dens).build()
    try:
        assert not g.walk_records()
        cam = osc.demo_camera(96, 64)
        assert g.generate_contacts_primary(as_capi_camera(cam)).tobytes() == og.trace_primary(cam)[0].tobytes()
        g.set_densities(2.0, 2.0)
        g.build()
        assert g.walk_records()
        assert g.generate_contacts_primary(as_capi_camera(cam)).tobytes() == oracle_tlg('cube')[1].trace_primary(cam)[0].tobytes()
    finally:
        g.close(); sc.close()


@pytest.mark.parametrize('name,res', [('gardenlight', (333, 251)), ('tenryuu', (640, 480))])
def test_generate_frame_equals_trace_then_shade(gpu_ctx, name, res):
    """rt_tlg_generate_frame (round 5): contacts and colours of a frame in ONE launch -- the same bytes as generateContacts(Camera&) followed by
    the shading pass (OpenCLRayTracerDemo.cpp:287-316), the contacts the oracle's bit for bit and the colours within 1e-5 relative of its
    cpuShadeBlinnPhong restatement (tolerance of north_star)."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, og = oracle_tlg(name)
    sc, gg = gpu_tlg(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    ccam = as_capi_camera(cam)
    npx = res[0] * res[1]
    want_c, _ = og.trace_primary(cam)
    want = orc.shade_frame(osc, cam, want_c)
    dc, dr, dc2, dr2 = gpu_ctx.alloc(npx * 32), gpu_ctx.alloc(npx * 16), gpu_ctx.alloc(npx * 32), gpu_ctx.alloc(npx * 16)
    try:
        gpu_ctx.fill_u32(dr, 0x7FC00000, npx * 4); gpu_ctx.fill_u32(dc, 0xDEADBEEF, npx * 8)
        gg.generate_frame_d(ccam, dc, dr)
        gg.generate_contacts_primary_d(ccam, dc2)
        capi.shade_blinn_phong_frame_d(gpu_ctx, sc, ccam, dc2, dr2)
        gpu_ctx.synchronize()
        got_c = gpu_ctx.download(dc, npx, capi.CONTACT)
        got = gpu_ctx.download(dr, npx * 4, np.float32).reshape(npx, 4)
        assert got_c.tobytes() == want_c.tobytes()
        assert got_c.tobytes() == gpu_ctx.download(dc2, npx, capi.CONTACT).tobytes()
        assert got.tobytes() == gpu_ctx.download(dr2, npx * 4, np.float32).tobytes()
        err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        assert float(np.where(want == got, 0.0, err).max()) <= 1e-5
    finally:
        for d in (dc, dr, dc2, dr2):
            gpu_ctx.free(d)


def test_packet_walk_cameras_that_move_turn_and_look_along_axes(gpu_ctx):
    """The packet walk (round 5) takes waves whose rays share their direction signs and carries, per camera position, the planes of the grid
    relative to the origin (written together with the camera-relative triangles when the camera moves or the grid is rebuilt).  Cameras that
    move and turn between frames, a camera looking straight along an axis (the image's centre lines are waves of mixed signs and of zero
    components: they walk lane by lane or with the reference's divisions), a rebuild between two frames of the same camera: every frame the
    oracle's bit for bit."""
    from openclraytracer_amd import capi
    name = 'gardenlight'
    osc, og = oracle_tlg(name)
    sc = capi.DeviceScene(gpu_ctx, packed_scene(name))
    g = capi.DeviceTLG(gpu_ctx, sc).build()
    try:
        assert g.walk_records()
        base = osc.demo_camera(200, 152)
        cams = []
        for k in range(6):
            c = base.copy()
            c['m'][0][3] += np.float32(0.37 * k); c['m'][0][7] -= np.float32(0.21 * k); c['m'][0][11] += np.float32(0.13 * k)     # (3 x 4, row-major: the last column is the position)
            cams.append(c)
        # looking along -z / +x with an axis-aligned frame: rows and columns through the centre have zero direction components
        for rot in (np.eye(3, dtype=np.float32), np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float32)):
            c = base.copy()
            m = c['m'][0].reshape(3, 4)
            m[:3, :3] = rot
            cams.append(c)
        for round_ in range(2):
            for k, cam in enumerate(cams):
                want, _ = og.trace_primary(cam)
                got = g.generate_contacts_primary(as_capi_camera(cam))
                assert got.tobytes() == want.tobytes(), (round_, k)
                if k == 2:
                    g.build()                              # the grid's records are rewritten: the camera's tables follow
                    again = g.generate_contacts_primary(as_capi_camera(cam))
                    assert again.tobytes() == want.tobytes(), (round_, k, 'after the rebuild')
    finally:
        g.close(); sc.close()


@pytest.mark.parametrize('name', ['gardenlight', 'tenryuu'])
def test_packet_walk_of_coherent_ray_buffers(gpu_ctx, name):
    """generateContacts(CLBuffer& rays, ..) through the grid with rays that travel together (round 5: waves of a ray buffer whose rays share their
    direction signs take the packet walk too, with the planes of the global table and the scene's triangle table, every lane its own origin): the
    camera's rays as Ray[] in row order and in 8x8 blocks, and a sheet of parallel rays from origins spread over a plane (the shape of shadow
    rays).  Every contact the oracle's."""
    import oracle as orc
    from openclraytracer_amd import capi
    osc, og = oracle_tlg(name)
    _, gg = gpu_tlg(gpu_ctx, name)
    cam = osc.demo_camera(256, 192)
    rays = orc.generate_rays(cam)
    want, st = og.trace_rays(rays)
    assert st['hits'] > 100
    assert gg.generate_contacts_rays(rays.view(capi.RAY)).tobytes() == want.tobytes()
    # 8x8 blocks: 64 consecutive rays = one block of pixels
    idx = np.arange(256 * 192).reshape(192 // 8, 8, 256 // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
    blocks = np.ascontiguousarray(rays[idx])
    want_b, _ = og.trace_rays(blocks)
    assert gg.generate_contacts_rays(blocks.view(capi.RAY)).tobytes() == want_b.tobytes()
    # parallel rays from a sheet of origins
    lo, hi = og.grid['lo'][0][:3], og.grid['hi'][0][:3]
    n = 64 * 300
    sheet = np.zeros(n, dtype=orc.RAY)
    sheet['idx'] = np.arange(n)
    u, v = np.meshgrid(np.linspace(0.0, 1.0, 300, dtype=np.float32), np.linspace(0.0, 1.0, 64, dtype=np.float32), indexing='ij')
    sheet['origin'][:, 0] = (lo[0] + (hi[0] - lo[0]) * u.reshape(-1)).astype(np.float32)
    sheet['origin'][:, 1] = np.float32(hi[1] + 0.25 * (hi[1] - lo[1]))
    sheet['origin'][:, 2] = (lo[2] + (hi[2] - lo[2]) * v.reshape(-1)).astype(np.float32)
    d = np.array([0.21, -0.95, 0.23], dtype=np.float32)
    sheet['direction'][:, :3] = d / np.float32(np.linalg.norm(d))
    want_s, st_s = og.trace_rays(sheet)
    assert st_s['hits'] > 50
    assert gg.generate_contacts_rays(sheet.view(capi.RAY)).tobytes() == want_s.tobytes()
