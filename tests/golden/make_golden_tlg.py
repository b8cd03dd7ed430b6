#!/usr/bin/env python3
"""Writes tests/golden/tlg.json: known answers of the Two-Level-Grid ORACLE (oracle/tlg_oracle.c) for the five
bundled models -- grid data, sizes of every build product, SHA-256 of the products, hit counts and a sum of t.

    python tests/golden/make_golden_tlg.py

The oracle is a restatement (parity unpinned, oracle/tlg_oracle.h); the fixture pins the oracle against regressions and
lets the `-m "not gpu"` suite check it without recomputing the expected values from the same code path alone: the
Gardenlight 64x48 hit count is the one number SURVEY.md recorded from the reference's own headers (reference_probe.json).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def sha(a):
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    import oracle as orc
    from conftest import MODELS, packed_scene
    out = {}
    for name in MODELS:
        sc = orc.Scene(packed_scene(name))
        g = orc.TLG(sc)
        e = dict(info=g.info, grid=dict(res=[int(x) for x in g.grid['res'][0]], step_hex=[float(x).hex() for x in g.grid['step'][0]]),
                 sha=dict(grid=sha(g.grid), top_cells=sha(g.top_cells), top_pairs=sha(g.top_pairs), cell_ranges=sha(g.cell_ranges),
                          leaf_pairs=sha(g.leaf_pairs), leaf_ranges=sha(g.leaf_ranges)), frames={})
        for res in ((64, 48), (640, 480)):
            cam = sc.demo_camera(*res)
            c, st = g.trace_primary(cam)
            e['frames']['%dx%d' % res] = dict(stats={k: (v if k != 'sum_t' else float(v).hex()) for k, v in st.items()}, contacts_sha=sha(c))
        out[name] = e
    json.dump(out, open(os.path.join(HERE, 'tlg.json'), 'w'), indent=1, sort_keys=True)
    print('wrote tlg.json')


if __name__ == '__main__':
    main()
