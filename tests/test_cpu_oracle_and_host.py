"""CPU suite (no GPU): the oracle against the committed fixtures and the reference-derived known answers,
the loaders, and the C-ABI library's exported surface."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import MODELS, ROOT, oracle_bvh, packed_scene

GOLD = os.path.join(ROOT, 'tests', 'golden')


def gold(name):
    return json.load(open(os.path.join(GOLD, name)))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ---------------------------------------------------------------- loaders
@pytest.mark.parametrize('name', MODELS)
def test_scene_pack_matches_fixture(name):
    buf = packed_scene(name)
    g = gold('scenes.json')[name]
    assert len(buf) == g['bytes'] and sha(buf) == g['sha256']
    probe = gold('reference_probe.json')
    import scene_pack
    assert scene_pack.scene_header(buf)['totalNumberOfTriangles'] == probe['triangles'][name]


def test_obj_parser_matches_reference_tinyobj_when_available(tmp_path):
    """Where oracle/_ref/tinyobj_dump (the reference's own parser, built from /root/reference) exists, the
    Python restatement must equal it field by field; its dump hashes are the committed fixture."""
    import scene_pack
    import subprocess
    dumper = os.path.join(ROOT, 'oracle', '_ref', 'tinyobj_dump')
    if not os.path.exists(dumper):
        pytest.skip('oracle/_ref not built (no /root/reference on this box)')
    g = gold('tinyobj.json')
    models = {'cube': 'assets/testmodels/Cube/cube.obj', 'gardenlight': 'assets/testmodels/Gardenlight/gardenlight01.obj',
              'tenryuu': 'assets/testmodels/Tenryuu/light-cruiser-tenryuu.obj'}
    for name, obj in models.items():
        out = str(tmp_path / ('%s.bin' % name))
        subprocess.check_call([dumper, os.path.join(ROOT, obj), out])
        assert sha(open(out, 'rb').read()) == g[name]['sha256']
        rs, rm = scene_pack.load_tinyobj_dump(out)
        ms, mm = scene_pack.parse_obj(os.path.join(ROOT, obj))
        assert len(rs) == len(ms) and len(rm) == len(mm)
        for a, b in zip(ms, rs):
            assert a.positions.tobytes() == b.positions.tobytes()
            assert a.indices.tobytes() == b.indices.tobytes()
            assert a.material_ids.tobytes() == b.material_ids.tobytes()
        for a, b in zip(mm, rm):
            assert a.as_floats().tobytes() == b.as_floats().tobytes()


def test_tinyobj_float_parser_quirks():
    import scene_pack
    ok, v = scene_pack.try_parse_double('0.1`')
    assert ok and np.float32(v) == np.float32(0.1)
    assert scene_pack.try_parse_double('-.5')[0] is False          # needs a digit before the dot
    assert scene_pack.try_parse_double('1e')[0] is False
    ok, v = scene_pack.try_parse_double('11e2')
    assert ok and v == 1100.0
    ok, v = scene_pack.try_parse_double('-0')
    assert ok and v == 0.0


def test_scene_file_edge_cases(tmp_path):
    """empty scene, unknown keys, light-only scene, bad light values (Scene.cpp:233-346)."""
    import scene_pack
    p = tmp_path / 's.txt'
    p.write_text('FOO bar\nLIGHT 1.0 2.0 3.0 4.0\nLIGHT x 2 3 4\n\nnokey\n')
    raw = scene_pack.parse_scene_file(str(p))
    assert sorted(raw) == ['FOO', 'LIGHT']
    buf = scene_pack.pack_scene(raw, [])
    h = scene_pack.scene_header(buf)
    assert h['numberOfLights'] == 2 and h['numberOfModels'] == 0 and h['totalNumberOfTriangles'] == 0
    lights = np.frombuffer(buf, dtype='<f4', count=8, offset=96).reshape(2, 4)
    assert lights[0].tolist() == [1.0, 2.0, 3.0, 4.0]
    assert lights[1].tolist() == [0.0, 0.0, 0.0, 0.0]               # invalid light line stays zero
    assert h['totalDataSize'] == 96 + 32


def _host_tools():
    import subprocess
    tool = os.environ.get('RTMI_SCENE_PACK_TOOL') or os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')      # (the sanitizer build: tools/san)
    if not os.path.exists(tool):
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'openclraytracer_amd', 'csrc', 'host'),
                               os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')])
    return tool


@pytest.mark.parametrize('name', MODELS)
def test_cpp_scene_loader_equals_oracle_packer(name, tmp_path):
    """the PRODUCT's C++ Scene::load packing (SceneLoader.cpp) == the oracle's packer == the committed hash"""
    import subprocess
    out = str(tmp_path / 's.bin')
    subprocess.check_call([_host_tools(), os.path.join('assets', 'scenes', name + '.txt'), out], cwd=ROOT)
    data = open(out, 'rb').read()
    assert data == packed_scene(name)
    assert sha(data) == gold('scenes.json')[name]['sha256']


def test_cpp_scene_loader_number_grammar(tmp_path):
    """the number grammar of the reference's OBJ parser (tiny_obj_loader.h:266-383): the C++ loader and the oracle packer
    must agree byte for byte on tokens a strtod would read differently -- and the parser stage of the oracle is pinned on
    the reference's own tinyobj (oracle/_ref/tinyobj_dump, test_obj_parser_equals_reference_tinyobj)"""
    import subprocess
    import scene_pack
    odd = ['1e', '-.5', '11e2', '0.1`', '2.5E-1', '+3.', '1.e1', '1e+', '007.250', '-0', '1.0000001', '12345678.9', '1e-3',
           '3.4028235e38', '1.17549435e-38', '0.30000000000000004', '9.999999e-1', '.', '+', '1e-', '5x', '6.02E+2z']
    lines = ['v %s %s %s' % (odd[i], odd[(i + 1) % len(odd)], odd[(i + 2) % len(odd)]) for i in range(len(odd))]
    lines += ['f %d %d %d' % (i + 1, (i + 1) % len(odd) + 1, (i + 2) % len(odd) + 1) for i in range(len(odd))]
    (tmp_path / 'odd.obj').write_text('\n'.join(lines) + '\n')
    (tmp_path / 's.txt').write_text('LIGHT 0.0 10.0 -330.0 1000.0\nMESH odd.obj\n')
    out = str(tmp_path / 's.bin')
    subprocess.check_call([_host_tools(), 's.txt', out], cwd=str(tmp_path))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        want = scene_pack.load_scene('s.txt')
    finally:
        os.chdir(cwd)
    assert open(out, 'rb').read() == want
    assert scene_pack.scene_header(want)['totalNumberOfTriangles'] == len(odd)


def test_cpp_scene_loader_error_behaviour(tmp_path):
    import subprocess
    r = subprocess.run([_host_tools(), str(tmp_path / 'missing.txt'), str(tmp_path / 'o.bin')], capture_output=True, text=True)
    assert r.returncode == 1 and "couldn't be opened" in r.stderr
    p = tmp_path / 's.txt'
    p.write_text('LIGHT 0.0 10.0 -330.0 1000.0\nMESH nowhere/none.obj')
    r = subprocess.run([_host_tools(), str(p), str(tmp_path / 'o.bin')], capture_output=True, text=True)
    assert r.returncode == 1 and 'Cannot open file' in r.stderr


# ---------------------------------------------------------------- oracle vs fixtures
def test_struct_sizes_match_reference_probe():
    import oracle as orc
    for k, v in gold('reference_probe.json')['struct_sizes'].items():
        got = orc.lib().orc_sizeof(k.encode())
        if got:
            assert got == v, k


@pytest.mark.parametrize('name', MODELS)
def test_bvh_fixture_and_invariants(name):
    """hashes + the invariants of Include/Testing/BVHTest.h and sortingTest.h."""
    _, b = oracle_bvh(name)
    g = gold('bvh.json')[name]
    assert b.n == g['leaves'] and b.N == g['morton_items']
    assert sha(b.pairs.tobytes()) == g['pairs_sha256'] and sha(b.nodes.tobytes()) == g['nodes_sha256']
    rc, depth = b.check_hierarchy()
    assert rc == 0 and depth == g['depth']
    assert depth + 1 == gold('reference_probe.json')['tree_depth_nodes'][name]
    keys = b.pairs['key']
    assert (np.diff(keys.astype(np.int64)) >= 0).all()                       # sorted
    assert sorted(b.pairs['value'][:b.n].tolist()) == list(range(b.n))        # permutation of the leaves
    assert (b.pairs['key'][b.n:] == 0xFFFFFFFF).all()                         # padding stays at the end
    # parent box contains child boxes (testBoundingBoxCorrectness)
    nodes = b.nodes
    inner = np.arange(b.n, 2 * b.n - 1)
    for side in (1, 2):
        ch = nodes['data'][inner, side]
        assert (nodes['lo'][inner, :3] <= nodes['lo'][ch, :3]).all()
        assert (nodes['hi'][inner, :3] >= nodes['hi'][ch, :3]).all()


@pytest.mark.parametrize('name', MODELS)
def test_dfs_leaf_order_is_sorted_pair_order(name):
    """the fact the HIP kernels' tie rule rests on: A-first DFS visits leaves in sorted-array order."""
    _, b = oracle_bvh(name)
    n, data = b.n, b.nodes['data']
    order, stack = [], [n]
    while stack:
        c = stack.pop()
        if c < n:
            order.append(c)
        else:
            stack.append(int(data[c][2]))
            stack.append(int(data[c][1]))
    assert np.array_equal(np.array(order, dtype=np.uint32), b.pairs['value'][:n])


@pytest.mark.parametrize('name', MODELS)
def test_trace_matches_fixture_and_reference_probe(name):
    osc, b = oracle_bvh(name)
    fx = gold('frames.json')[name]
    probe = gold('reference_probe.json')
    for key in ('64x48', '640x480'):
        w, h = map(int, key.split('x'))
        cam = osc.demo_camera(w, h)
        assert [int(x) for x in cam.view(np.uint32).ravel()] == fx[key]['camera_bits']
        c, st = b.trace_primary(cam)
        g = fx[key]
        assert sha(c.tobytes()) == g['contacts_sha256']
        assert (st.hits, st.inner_visits, st.leaf_tests, st.max_stack) == (g['hits'], g['inner_visits'],
                                                                           g['leaf_tests'], g['max_stack'])
        if key == '640x480':     # numbers produced by the reference's own headers during the survey
            assert round(st.inner_visits / len(c), 1) == probe['inner_visits_per_ray_640x480'][name]
            assert round(st.leaf_tests / len(c), 1) == probe['leaf_tests_per_ray_640x480'][name]
            assert st.max_stack == probe['max_stack_640x480'][name]
        elif name in probe['hits_64x48_fixed']:
            assert st.hits == probe['hits_64x48_fixed'][name]


@pytest.mark.parametrize('name,res', [('cube', (64, 48)), ('gardenlight', (64, 48)), ('tenryuu', (40, 30))])
def test_bvh_equals_brute_force(name, res):
    """FIXED-mode property recorded by the survey for the reference: BVH contacts == brute force, t bit-exact."""
    import oracle as orc
    osc, b = oracle_bvh(name)
    cam = osc.demo_camera(*res)
    c, _ = b.trace_primary(cam)
    br = orc.brute_primary(osc, cam)
    assert np.array_equal(c['normal_t'][:, 3].view(np.uint32), br['normal_t'][:, 3].view(np.uint32))


def test_small_contact_and_rgba_fixtures():
    import oracle as orc
    z = np.load(os.path.join(GOLD, 'contacts_64x48.npz'))
    for name in MODELS:
        osc, b = oracle_bvh(name)
        cam = osc.demo_camera(64, 48)
        c, _ = b.trace_primary(cam)
        assert np.array_equal(c.view(np.uint8).reshape(-1, 32), z[name + '_contacts'])
        rgba = orc.shade_frame(osc, cam, c)
        want = z[name + '_rgba']
        assert np.allclose(rgba, want, rtol=1e-5, atol=0)
        miss = c['normal_t'][:, 3] <= 0
        assert (rgba[miss] == 0).all() and (rgba[~miss][:, 3] == 1).all()


def test_trace_rays_equals_primary_and_handles_edge_cases():
    import oracle as orc
    osc, b = oracle_bvh('gardenlight')
    cam = osc.demo_camera(32, 24)
    rays = orc.generate_rays(cam)
    c1, _ = b.trace_primary(cam)
    c2, _ = b.trace_rays(rays)
    assert c1.tobytes() == c2.tobytes()
    empty, _ = b.trace_rays(rays[:0])
    assert len(empty) == 0
    # axis-parallel and degenerate directions (1/0 = inf, 0*inf = NaN paths of AABB.h:126-142)
    odd = rays[:4].copy()
    odd['direction'][0] = (0, 0, 1, 0)
    odd['direction'][1] = (0, 1, 0, 0)
    odd['direction'][2] = (0, 0, 0, 0)
    odd['direction'][3] = (np.nan, 0, 1, 0)
    c3, _ = b.trace_rays(odd)
    assert np.isfinite(c3['normal_t'][:, 3]).all()


def test_kat_scan_and_sort():
    import oracle as orc
    kat = gold('kat.json')
    rng = np.random.default_rng(1234)
    for n in [2, 64, 1024, 65536]:
        a = rng.integers(0, 1 << 10, size=n, dtype=np.uint32)
        s = orc.scan_inclusive(a)
        assert np.array_equal(s, np.cumsum(a, dtype=np.uint64).astype(np.uint32))          # ps[i] == in[i] + ps[i-1]
        assert sha(s.tobytes()) == kat['scan_%d' % n]
        pairs = np.zeros(n, dtype=orc.PAIR)
        pairs['key'] = rng.integers(0, 16, size=n, dtype=np.uint32)
        pairs['value'] = np.arange(n, dtype=np.uint32)
        out = orc.bitonic_sort(pairs)
        assert (np.diff(out['key'].astype(np.int64)) >= 0).all()
        assert sorted(out['value'].tolist()) == list(range(n))                              # key/value integrity
        assert sha(out.tobytes()) == kat['sort_%d' % n]


# ---------------------------------------------------------------- the C-ABI library (no compute without a GPU)
def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rtmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rt_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    from openclraytracer_amd import capi
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    from openclraytracer_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        pytest.fail('librtmi.so is not built: run __graft_entry__.build()')
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert lib.rt_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    """On a box without a gfx950 device context creation must fail loudly (no fallback path)."""
    import torch
    from openclraytracer_amd import capi
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(capi.RtError):
        capi.Context(0)


def test_tile_slot_map_is_a_bijection():
    from openclraytracer_amd import capi
    for (w, h, n) in [(1920, 1080, 8), (640, 480, 2), (65, 64, 3), (64, 64, 1)]:
        rank, slot = capi.tile_slot_map(w, h, n)
        cam = np.zeros(1, dtype=capi.CAMERA)
        cam['resX'], cam['resY'] = w, h
        seen = set()
        for r in range(n):
            owned = capi.tiles_owned_pixels(cam, r, n)
            s = slot[rank == r]
            assert s.size == 0 or s.max() < owned
            assert len(set(s.tolist())) == s.size
            seen.update((r, int(x)) for x in s)
        assert len(seen) == w * h


def test_synthetic_scene_loaders_agree(tmp_path):
    """Bunny-class stand-in (quads, 'o' groups, 35 k vertices in one shape): C++ loader == Python packer."""
    import subprocess
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
    import gen_scenes
    import scene_pack
    txt = gen_scenes.ensure()['synth_bunny']
    out = str(tmp_path / 'b.bin')
    subprocess.check_call([_host_tools(), os.path.relpath(txt, ROOT), out], cwd=ROOT)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        want = scene_pack.load_scene(txt)
    finally:
        os.chdir(cwd)
    assert open(out, 'rb').read() == want
    assert scene_pack.scene_header(want)['totalNumberOfTriangles'] == 69564


def test_literal_zero_hit_counts():
    """LITERAL_ZERO = the reference's guard as written (inner node n-2 never built) AND its 64-wide refit launch, whose work
    items beyond the last leaf are clamped to it (BVHKernels.cl:67, BVHManager.cpp:213-217): those extra walkers re-merge the
    ancestors of the last leaf, which repairs part of the spine above the lost node.  SURVEY.md 8(c) quotes 1552 hits for Cube
    64x48: that is the same tree refitted WITHOUT the clamped work items (one walker per leaf), reproduced here."""
    import oracle as orc
    want = {'cube': (2209, 2209, 1552), 'gardenlight': (624, 624, 272), 'man': (1580, 1513, 434), 'clumsydragon': (533, 257, 257)}
    for name, (fixed, literal, one_walker_per_leaf) in want.items():
        sc = orc.Scene(packed_scene(name))
        cam = sc.demo_camera(64, 48)
        hits = lambda b: int((b.trace_primary(cam)[0]['normal_t'][:, 3] > 0).sum())
        assert hits(orc.BVH(sc, 0)) == fixed
        b = orc.BVH(sc, 1)
        assert hits(b) == literal
        nodes = orc.build_stages(sc, 1)['radix'].copy()
        orc.lib().orc_refit(orc._p(nodes), sc.n_tris, 0)          # n walkers, no clamped extras
        b.nodes = nodes
        assert hits(b) == one_walker_per_leaf


def test_bench_reads_the_tag_of_a_traversal_kernel_name():
    """rocprofv3 names the traversal kernels by their template arguments; bench.py and tools/summarize_profiles.py tell the
    steady-state launches (second argument 0), the calibration (1) and the isolated measurement pass (2) apart by them, whatever
    further arguments a kernel has"""
    import bench
    assert bench.kernel_tag('k_trace<1, 2>') == (1, 2)
    assert bench.kernel_tag('k_trace_packet4<1, 0, 1, 1, 0>') == (1, 0)
    assert bench.kernel_tag('k_trace_long<0, 1>') == (0, 1)
    assert bench.kernel_tag('k_shade') is None


def test_bench_summary_line_fits_a_stdout_tail():
    """The ONE line bench.py prints is a summary of at most 4 096 bytes (round 4's 20 KB line outgrew the tail of stdout the driver
    keeps, and its record of the run held no parsed line); the whole record goes to --details.  Checked on round 4's committed full
    record and on its in-process rehearsal of four ranks: contract keys, the two added objects, one figure per further leg."""
    import bench
    for src, tiled in (('r04_bench_line.json', False), ('r04_bench_group4_rehearsal.json', True), ('r05_bench_details.json', False), ('r05_bench_group4_rehearsal.json', True)):
        full = json.loads(open(os.path.join(ROOT, 'profiles', src)).read())
        line = bench.compact_line(full, os.path.join(ROOT, 'gpurun_out', 'bench_details.json'))
        text = json.dumps(line)
        assert len(text) <= bench.LINE_LIMIT == 4096
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
            assert k in line and (k == 'data' or line[k] == full[k]), k
        assert line['config']['workload'] == full['config']['workload'] and 'model' not in line['config']
        rf = line['roofline']
        assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] == 'GB/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
        assert rf['traffic'] == full['roofline']['traffic'] and rf['frac_of_output_floor'] == full['roofline']['frac_of_output_floor']
        assert line['details_file'] == 'gpurun_out/bench_details.json'
        if tiled:
            assert line['cpu_baseline'] is None and len(line['per_rank']['rows']) == 4
        else:
            cb = line['cpu_baseline']
            assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['sample']
            assert line['verified']['frames'] == 5 and line['verified']['config4'] is True
            assert line['legs']['config4']['mrays'] == full['config4']['mrays_per_s_grid_kept']
            assert set(line['legs']['frame_ms']) == set(MODELS)


def test_host_cpu_shade_blinn_phong_equals_the_oracle():
    """the host mirror's cpuShadeBlinnPhong(Camera*, Contact*, char*, CL_FLOAT4&) (Include/CLData/Shading.h:74-166; what an unmodified
    RTFrame calls per contact, OpenCLRayTracerDemo.cpp:298-316) -- csrc/shade_math.h compiled by g++ -- gives the oracle's colours bit
    for bit on the oracle's contacts (same operations, same libm), on every bundled model.  No GPU involved."""
    import ctypes as C
    import oracle as orc
    lib = C.CDLL(os.environ.get('RTMI_HOST_LIB') or os.path.join(ROOT, 'openclraytracer_amd', 'librtmi_host.so'))
    fn = getattr(lib, '_Z18cpuShadeBlinnPhongP6CameraP7ContactPcR10CL_FLOAT4_')
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = None
    for name in MODELS:
        buf = packed_scene(name)
        osc = orc.Scene(buf)
        ob = orc.BVH(osc, 0)
        cam = osc.demo_camera(64, 48)
        contacts, _ = ob.trace_primary(cam)
        want = orc.shade_frame(osc, cam, contacts).reshape(-1, 4)
        scene = np.frombuffer(buf, dtype=np.uint8).copy()
        out = np.zeros(4, dtype=np.float32)
        hits = np.nonzero(contacts['normal_t'][:, 3] > 0)[0]
        assert len(hits) > 100
        for i in hits:
            c = contacts[i:i + 1].copy()
            fn(cam.ctypes.data, c.ctypes.data, scene.ctypes.data, out.ctypes.data)
            assert out.tobytes() == want[c['pixelIndex'][0]].tobytes(), (name, int(i))


def test_bench_verification_fixture_and_comparator():
    """tests/golden/bench_frames.json (what bench.py verifies its timed frames against): the default-camera records agree with
    frames.json, the oracle reproduces one of its moving-camera poses and one ray-buffer leg at a small cost, and bench.rgba_mismatch
    accepts the oracle's own colours, tolerates 1e-5 relative and rejects a wrong pixel, a wrong alpha and a shifted sum."""
    import bench
    import oracle as orc
    fx = bench.bench_fixture()
    frames = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'frames.json')))
    for name in MODELS:
        e = fx['scenes'][name]['1920x1080']
        assert e['default']['contacts_sha256'] == frames[name]['1920x1080']['contacts_sha256']
        assert e['default']['camera_bits'] == frames[name]['1920x1080']['camera_bits']
        assert e['ray_buffer']['primary']['contacts_sha256'] == e['default']['contacts_sha256']      # generateContacts2 on the camera's rays
        assert set(e) >= {'default', 'pose5', 'pose17', 'ray_buffer', 'scene_sha256'}
    assert fx['atrium']['3840x2160']['default']['rays'] == 3840 * 2160
    # one pose and the reflection leg of the cube, recomputed
    osc, ob = oracle_bvh('cube')
    e = fx['scenes']['cube']['1920x1080']
    cam = np.array(e['pose5']['camera_bits'], dtype=np.uint32).view(orc.CAMERA)
    c, st = ob.trace_primary(cam)
    assert hashlib.sha256(c.tobytes()).hexdigest() == e['pose5']['contacts_sha256'] and int(st.hits) == e['pose5']['hits']
    rgba = orc.shade_frame(osc, cam, c)
    assert bench.rgba_mismatch(rgba, e['pose5'], fx['sample_stride']) is None
    near = rgba * np.float32(1.0 + 4e-6)
    near[:, 3] = rgba[:, 3]
    assert bench.rgba_mismatch(near, e['pose5'], fx['sample_stride']) is None
    hit = int(np.nonzero(rgba[::fx['sample_stride'], 3] > 0)[0][0])
    bad = rgba.copy(); bad[fx['sample_stride'] * hit, 0] *= np.float32(0.99)
    assert 'sampled pixel' in bench.rgba_mismatch(bad, e['pose5'], fx['sample_stride'])
    bad = rgba.copy(); bad[fx['sample_stride'] * 5, 3] = 1.0 - bad[fx['sample_stride'] * 5, 3]
    assert bench.rgba_mismatch(bad, e['pose5'], fx['sample_stride']) is not None
    bad = rgba.copy(); bad[1::2, 0] *= np.float32(1.001)
    assert 'sums' in bench.rgba_mismatch(bad, e['pose5'], fx['sample_stride'])
    cam0 = osc.demo_camera(1920, 1080)
    c0, _ = ob.trace_primary(cam0)
    rays = orc.generate_secondary_rays(c0, 1, eps=fx['eps'], rays_in=orc.generate_rays(cam0))
    assert len(rays) == e['ray_buffer']['reflect']['rays'] and hashlib.sha256(rays.tobytes()).hexdigest() == e['ray_buffer']['reflect']['rays_sha256']
    cr, _ = ob.trace_rays(rays)
    assert hashlib.sha256(cr.tobytes()).hexdigest() == e['ray_buffer']['reflect']['contacts_sha256']
