#!/usr/bin/env python3
"""Writes tests/golden/search_tree_frames.json: inner-node visits per 1920x1080 frame when primary rays walk the SEARCH
tree of librtmi.so (openclraytracer_amd/csrc/build_kernels.hip, section 6) instead of the reference's tree.

    python tests/golden/make_golden_search.py

The search tree is restated here with the oracle's own building blocks: keys as k_morton_leaves computes them (fp32; search_tree_key, key_layout),
the reference's bitonic network and radix-tree restatements (oracle/rt_oracle.c), inner boxes as plain unions (what k_refit<TIGHT> computes), then the oracle's traversal with
its visit counters.  The leaf-test count and the contacts are those of the reference tree (asserted below: same reached
leaves); only `inner_visits` differs.  bench.py uses the numbers for the `walked` fields of its roofline object.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def expand_bits(v):
    v = v.astype(np.uint32)
    for mul, mask in ((0x00010001, 0xFF0000FF), (0x00000101, 0x0F00F00F), (0x00000011, 0xC30C30C3), (0x00000005, 0x49249249)):
        v = (v * np.uint32(mul)) & np.uint32(mask)
    return v


def search_tree(orc, b, scene_box):
    """(nodes[2n-1], depth) of the search tree over the leaves of oracle BVH b"""
    L = orc.lib()
    L.orc_radix_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.orc_refit.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    n = b.n
    f = np.float32
    lo, hi = b.nodes['lo'][:n, :3], b.nodes['hi'][:n, :3]
    rlo, rhi = scene_box                                                  # the scene header's box (bytes 64 / 80 of the packed scene), as the library
    with np.errstate(all='ignore'):
        c = (lo + hi) * f(0.5)
        v = (c - rlo) / (rhi - rlo)
        v = np.where(v == v, v, f(0.0))
    # the bits of the key: per level the axis along which the cell is longest (build_kernels.hip, key_layout); a box that stays within two bits
    # between its axes keeps ten bits per axis, x, y, z interleaved
    cell = np.maximum(rhi.astype(np.float64) - rlo.astype(np.float64), 0.0)
    seq, nbits = [], [0, 0, 0]
    for _ in range(30):
        a = 0
        if cell[1] > cell[a]: a = 1
        if cell[2] > cell[a]: a = 2
        seq.append(a); cell[a] *= 0.5; nbits[a] += 1
    if max(nbits) - min(nbits) >= 3:
        with np.errstate(all='ignore'):
            q = [np.minimum(np.minimum(np.maximum(v[:, k] * f(1 << nbits[k]), f(0.0)), f(1 << nbits[k])).astype(np.uint32), np.uint32((1 << nbits[k]) - 1)) for k in range(3)]
        left = list(nbits)
        key = np.zeros(n, dtype=np.uint32)
        for a in seq:
            left[a] -= 1
            key = (key << np.uint32(1)) | ((q[a] >> np.uint32(left[a])) & np.uint32(1))
    else:
        with np.errstate(all='ignore'):
            x = np.minimum(np.maximum(v * f(1024.0), f(0.0)), f(1023.0)).astype(np.uint32)
        key = expand_bits(x[:, 0]) * np.uint32(4) + expand_bits(x[:, 1]) * np.uint32(2) + expand_bits(x[:, 2])
    pairs = np.full(b.N, 0xFFFFFFFF, dtype=np.uint32).repeat(2).view(orc.PAIR).copy()
    pairs['key'][:n] = key
    pairs['value'][:n] = np.arange(n, dtype=np.uint32)
    pairs = orc.bitonic_sort(pairs)
    nodes = np.zeros(2 * n - 1, dtype=orc.NODE)
    nodes[:n] = b.nodes[:n]
    L.orc_radix_tree(nodes.ctypes.data, pairs.ctypes.data, n, 0)
    L.orc_refit(nodes.ctypes.data, n, 0)                                 # parents, depth; boxes as the REFERENCE refit makes them ...
    # ... and since round 3 the search tree's inner boxes are the plain unions of their children's (k_refit<TIGHT>, build_kernels.hip:
    # no maximum pulled up to FLT_MIN).  Bottom-up over the finished tree: a node's children are done before it in a post-order walk.
    data = nodes['data']
    order, stack = [], [n]
    while stack:
        i = stack.pop()
        if i >= n:
            order.append(i)
            stack.append(int(data[i, 1])); stack.append(int(data[i, 2]))
    for i in reversed(order):
        a, c = int(data[i, 1]), int(data[i, 2])
        nodes['lo'][i, :3] = np.minimum(nodes['lo'][a, :3], nodes['lo'][c, :3])
        nodes['hi'][i, :3] = np.maximum(nodes['hi'][a, :3], nodes['hi'][c, :3])
    d = C.c_uint32(0)
    L.orc_check_hierarchy(nodes.ctypes.data, n, C.byref(d))
    return nodes, d.value


def main():
    import oracle as orc
    from conftest import MODELS, packed_scene
    L = orc.lib()
    out = {}
    for name in MODELS:
        sc = orc.Scene(packed_scene(name))
        b = orc.BVH(sc, 0)
        hdr = np.frombuffer(sc.bytes, dtype=np.float32, count=24, offset=0)
        nodes, depth = search_tree(orc, b, (hdr[16:19].copy(), hdr[20:23].copy()))
        cam = sc.demo_camera(1920, 1080)
        ref, st_ref = b.trace_primary(cam)
        got = np.zeros(1920 * 1080, dtype=orc.CONTACT)
        st = orc.Stats()
        L.orc_trace_primary(cam.ctypes.data, nodes.ctypes.data, b.n, sc.arr.ctypes.data, got.ctypes.data, C.byref(st), orc.ncpu())
        assert st.leaf_tests == st_ref.leaf_tests and st.hits == st_ref.hits
        assert (got['normal_t'][:, 3] == ref['normal_t'][:, 3]).all()
        out[name] = {'1920x1080': dict(rays=1920 * 1080, inner_visits=int(st.inner_visits), leaf_tests=int(st.leaf_tests),
                                       reference_inner_visits=int(st_ref.inner_visits), depth=depth)}
        print(name, out[name])
    json.dump(out, open(os.path.join(HERE, 'search_tree_frames.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
