"""DESIGN.md section 2, "the first axis": a `0 * inf` in the reference's slab test is harmless on x and hierarchy-dependent on y / z.

The reference's staged test (AABB.h:126-142) is restated in numpy with C's NaN behaviour (tests/plane_rays.py) and the leaves a
ray reaches are computed (a) in the oracle's reference tree under that test and (b) in the SEARCH tree under the test the four-wide
packet kernel evaluates on its per-octant records (max3 / min3 ignoring NaN).  For rays whose only NaN sits on x the two sets are
equal on every sampled ray; for y and z the same comparison finds rays where they differ -- which is why only those rays are
sent to the reference tree (trace_kernels.hip, k_trace_packet4)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

import oracle as orc
import plane_rays as pr
from conftest import packed_scene

f32 = np.float32


def reached_sets(name, axis, n_planes, stride, seed=5):
    import make_golden_search as mgs
    sc = orc.Scene(packed_scene(name))
    b = orc.BVH(sc, 0)
    n = b.n
    hdr = np.frombuffer(sc.bytes, dtype=np.float32, count=24, offset=0)          # the scene header's box: bytes 64 / 80
    snodes, _ = mgs.search_tree(orc, b, (hdr[16:19].copy(), hdr[20:23].copy()))
    lo, hi = b.nodes['lo'][:, :3], b.nodes['hi'][:, :3]
    slo, shi = snodes['lo'][:, :3], snodes['hi'][:, :3]
    rng = np.random.default_rng(seed)
    rays_seen = with_nan = differ = 0
    for _ in range(n_planes):
        pl = b.nodes[('lo', 'hi')[int(rng.integers(0, 2))]][int(rng.integers(0, n)), axis]
        if pl == 0:
            continue
        cam = pr.plane_camera(sc, (64, 48), axis, pl)
        rays = orc.generate_rays(cam)
        with np.errstate(all='ignore'):
            inv = f32(1.0) / rays['direction'][:, :3]
        risky = np.nonzero(~np.isfinite(inv[:, axis]))[0]
        for r in risky[::stride]:
            o, iv = rays['origin'][r, :3], inv[r]
            ref = pr.reached_leaves(b.nodes, n, pr.staged_valid(lo, hi, o, iv))
            got = pr.reached_leaves(snodes, n, pr.octant_valid(slo, shi, o, iv))
            rays_seen += 1
            with np.errstate(all='ignore'):
                with_nan += int(np.isnan((lo[:, axis] - o[axis]) * iv[axis]).any())
            differ += int(not np.array_equal(ref, got))
    return rays_seen, with_nan, differ


@pytest.mark.parametrize('name', ['man', 'clumsydragon'])
def test_nan_on_the_first_axis_is_absorbed(name):
    rays, with_nan, differ = reached_sets(name, 0, 3, 4)
    assert rays >= 12 and with_nan == rays          # every sampled ray multiplies 0 by inf somewhere
    assert differ == 0


def test_nan_on_the_other_axes_depends_on_the_hierarchy():
    rays, with_nan, differ = reached_sets('man', 1, 4, 3)
    assert with_nan > 0 and differ > 0
