"""CPU tests of the Two-Level-Grid oracle (oracle/tlg_oracle.c) and of the host-side TLG logic of librtmi.so."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import MODELS, ROOT, packed_scene

GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tlg.json')))
PROBE = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_probe.json')))


def sha(a):
    return hashlib.sha256(a.tobytes()).hexdigest()


_tlg = {}


def oracle_tlg(name):
    import oracle as orc
    if name not in _tlg:
        sc = orc.Scene(packed_scene(name))
        _tlg[name] = (sc, orc.TLG(sc))
    return _tlg[name]


def test_reference_probe_gardenlight_hits():
    """the one TLG number SURVEY.md took from the reference's headers: Gardenlight 64x48 loses 48 of the BVH's 624 hits"""
    sc, g = oracle_tlg('gardenlight')
    c, st = g.trace_primary(sc.demo_camera(64, 48))
    p = PROBE['tlg_gardenlight_64x48']
    assert st['hits'] == p['bvh_hits'] - p['tlg_missed'] == int((c['normal_t'][:, 3] > 0).sum())


@pytest.mark.parametrize('name', MODELS)
def test_oracle_matches_golden_fixture(name):
    sc, g = oracle_tlg(name)
    e = GOLD[name]
    assert g.info == e['info']
    assert [int(x) for x in g.grid['res'][0]] == e['grid']['res']
    assert [float(x).hex() for x in g.grid['step'][0]] == e['grid']['step_hex']
    got = dict(grid=sha(g.grid), top_cells=sha(g.top_cells), top_pairs=sha(g.top_pairs), cell_ranges=sha(g.cell_ranges),
               leaf_pairs=sha(g.leaf_pairs), leaf_ranges=sha(g.leaf_ranges))
    assert got == e['sha']
    c, st = g.trace_primary(sc.demo_camera(64, 48))
    assert sha(c) == e['frames']['64x48']['contacts_sha']
    assert st['hits'] == e['frames']['64x48']['stats']['hits']


@pytest.mark.parametrize('name', MODELS)
def test_grid_structure_invariants(name):
    """ideas of Include/Testing/TwoLevelGridTest.h:54-160: pair counts, ranges partition the sorted arrays, every leaf
    pair lies in a leaf cell of a top cell that holds the triangle"""
    sc, g = oracle_tlg(name)
    i = g.info
    tp, lp = g.top_pairs, g.leaf_pairs
    assert (np.diff(tp['key'].astype(np.int64)) >= 0).all() and (np.diff(lp['key'].astype(np.int64)) >= 0).all()
    assert (tp['key'][:i['pairs_count']] < i['cells']).all() and (tp['key'][i['pairs_count']:] == 0xFFFFFFFF).all()
    w = i['written_leaf_pairs']
    assert (lp['key'][:w] < i['leaf_cells_count']).all() and (lp['key'][w:] == 0xFFFFFFFF).all()
    assert i['dropped_leaf_pairs'] == 0
    # ranges: [x, y) of every key, empty cells (0, 0)
    for pairs, ranges, count in ((tp, g.cell_ranges, i['pairs_count']), (lp, g.leaf_ranges, w)):
        keys = pairs['key'][:count]
        uniq, first, cnt = np.unique(keys, return_index=True, return_counts=True)
        want = np.zeros(len(ranges), dtype=ranges.dtype)
        want['key'][uniq] = first
        want['value'][uniq] = first + cnt
        assert ranges.tobytes() == want.tobytes()
    # leaf cells of a top cell are contiguous from firstLeafIdx, resolution product = count
    tc = g.top_cells
    per = tc['res'][:, 0].astype(np.int64) * tc['res'][:, 1] * tc['res'][:, 2]
    assert per.sum() == i['leaf_cells_count']
    assert (tc['firstLeafIdx'][1:] == np.cumsum(per)[:-1]).all() and tc['firstLeafIdx'][0] == 0
    # a leaf pair's triangle is paired with the enclosing top cell
    top_of_leaf = np.searchsorted(np.cumsum(per), lp['key'][:w], side='right')
    have = set(zip(tp['key'][:i['pairs_count']].tolist(), tp['value'][:i['pairs_count']].tolist()))
    sample = np.random.default_rng(5).choice(w, size=min(w, 2000), replace=False)
    assert all((int(top_of_leaf[k]), int(lp['value'][k])) in have for k in sample)


def test_pow_third_is_the_rounded_exact_power():
    """orc_pow_third == fp32 rounding of the double-precision x**fp32(1/3) (numpy's pow on float64 is accurate to < 1 ulp
    of double, far below fp32 rounding), and the C library's powf agrees except for its own last-ulp errors"""
    import oracle as orc
    rng = np.random.default_rng(11)
    xs = np.concatenate([np.exp(rng.uniform(-85, 85, 20000)), [1.0, 8.0, 27.0, 1e-45, 3.4e38, 2.0]]).astype(np.float32)
    y = float(np.float32(1.0) / np.float32(3.0))
    want = np.power(xs.astype(np.float64), y).astype(np.float32)
    got = np.array([orc.pow_third(float(x)) for x in xs], dtype=np.float32)
    assert got.tobytes() == want.tobytes()
    assert orc.pow_third(0.0) == 0.0 and np.isnan(orc.pow_third(-1.0)) and np.isinf(orc.pow_third(float('inf')))
    libm = ctypes.CDLL('libm.so.6')
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    ref = np.array([libm.powf(float(x), y) for x in xs], dtype=np.float32)
    ulp = np.abs(ref.view(np.int32).astype(np.int64) - got.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() > 0.99


def test_product_host_pow_third_equals_oracle():
    """librtmi.so's host-side tlg_pow_third (the same source the device kernels compile) == the oracle's, bit for bit"""
    import oracle as orc
    from openclraytracer_amd import capi
    rng = np.random.default_rng(12)
    xs = np.concatenate([np.exp(rng.uniform(-85, 85, 20000)), [0.0, 1.0, 8.0, 1e-45, 3.4e38]]).astype(np.float32)
    a = np.array([orc.pow_third(float(x)) for x in xs], dtype=np.float32)
    b = np.array([capi.tlg_pow_third(float(x)) for x in xs], dtype=np.float32)
    assert a.tobytes() == b.tobytes()


def test_triangle_box_test_known_cases():
    """AABBTriangleIntersect (AABB.h:477-540) on hand-checked configurations"""
    import oracle as orc
    L = orc.lib()
    F = orc.F4

    def sat(c, h, a, b, d):
        return bool(L.orc_aabb_triangle_intersect(F(*c, 0), F(*h, 0), F(*a, 0), F(*b, 0), F(*d, 0)))

    unit = ((0, 0, 0), (1, 1, 1))
    assert sat(*unit, (-0.5, -0.5, 0), (0.5, -0.5, 0), (0, 0.5, 0))              # inside
    assert sat(*unit, (-5, -5, 0.5), (5, -5, 0.5), (0, 5, 0.5))                  # big triangle cutting the box
    assert not sat(*unit, (-5, -5, 1.5), (5, -5, 1.5), (0, 5, 1.5))              # plane above the box
    assert not sat(*unit, (2, 2, 0), (3, 2, 0), (2, 3, 0))                       # far away
    assert not sat(*unit, (1.2, 1.2, -3), (1.2, 1.2, 3), (3, 3, 0))              # box of the triangle overlaps, triangle misses the corner
    assert sat(*unit, (1, -1, 0), (1, 1, 0), (2, 0, 0))                          # touching a face counts (<=)


def test_tlg_traversal_is_close_to_but_not_brute_force():
    """TLG contacts are a subset of the BVH's (t bit-identical where both hit the same triangle) -- the reference's grid
    loses a few hits (first leaf cell with any hit wins, resolution-0 cells are skipped); SURVEY.md section 8(c)"""
    import oracle as orc
    for name, res in (('cube', (64, 48)), ('tenryuu', (160, 120))):
        sc, g = oracle_tlg(name)
        cam = sc.demo_camera(*res)
        c, _ = g.trace_primary(cam)
        b, _ = orc.BVH(sc, 0).trace_primary(cam)
        hit = c['normal_t'][:, 3] > 0
        assert (b['normal_t'][:, 3][hit] > 0).all()                              # no hit the BVH does not have
        same = c['normal_t'][:, 3][hit] == b['normal_t'][:, 3][hit]
        assert same.mean() > 0.99
