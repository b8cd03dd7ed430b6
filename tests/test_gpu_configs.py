"""BASELINE.json configs 3, 4 and 5 on the seeded procedural stand-ins (the real Bunny / Sponza are not shipped with
the reference): full-size parity against the oracle through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, as_capi_camera

pytestmark = pytest.mark.gpu


def _synth_scene_bytes(name):
    sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
    import gen_scenes
    txt = gen_scenes.ensure()[name]
    tool = os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')
    out = '/tmp/_%s.bin' % name
    subprocess.check_call([tool, os.path.relpath(txt, ROOT), out], cwd=ROOT)      # the product's C++ loader
    return open(out, 'rb').read()


def test_config3_bunny_class_full_hd(gpu_ctx):
    """~70 k triangles, 1920x1080, BVH: every contact bit-exact; tree depth 2869 exercises the deferral path."""
    import oracle as orc
    from openclraytracer_amd import capi
    buf = _synth_scene_bytes('synth_bunny')
    osc = orc.Scene(buf)
    ob = orc.BVH(osc, 0)
    assert osc.n_tris == 69564
    sc = capi.DeviceScene(gpu_ctx, buf)
    gb = capi.DeviceBVH(gpu_ctx, sc).build()
    assert gb.read_nodes().tobytes() == ob.nodes.tobytes()
    cam = osc.demo_camera(1920, 1080)
    want, st = ob.trace_primary(cam)
    got = gb.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()
    assert st.hits > 300000
    gb.close(); sc.close()


def test_config4_atrium_class_two_level_grid_full_hd(gpu_ctx):
    """~262 k triangles, Two-Level Grid build (densities 2.0 / 2.0) + trace at 1920x1080: build products and every contact
    bit-exact against the oracle."""
    import oracle as orc
    from openclraytracer_amd import capi
    buf = _synth_scene_bytes('synth_atrium')
    osc = orc.Scene(buf)
    og = orc.TLG(osc)
    sc = capi.DeviceScene(gpu_ctx, buf)
    gg = capi.DeviceTLG(gpu_ctx, sc).build()
    info = gg.info()
    assert all(info[k] == og.info[k] for k in info)
    assert gg.read(gg.TOP_CELLS).tobytes() == og.top_cells.tobytes()
    assert gg.read(gg.LEAF_PAIRS).tobytes() == og.leaf_pairs.tobytes()
    assert gg.read(gg.LEAF_RANGES).tobytes() == og.leaf_ranges.tobytes()
    cam = osc.demo_camera(1920, 1080)
    want, st = og.trace_primary(cam)
    got = gg.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()
    assert st['hits'] > 500000
    gg.close(); sc.close()


def test_config5_atrium_class_4k_tiled_over_8_ranks(gpu_ctx):
    """~262 k triangles, 3840x2160, frame tiled over 8 ranks (run one after the other on this GPU): the ranks' compact
    buffers hold exactly the oracle's contacts of their tiles."""
    import oracle as orc
    from openclraytracer_amd import capi
    buf = _synth_scene_bytes('synth_atrium')
    osc = orc.Scene(buf)
    ob = orc.BVH(osc, 0)
    assert osc.n_tris == 262656
    sc = capi.DeviceScene(gpu_ctx, buf)
    gb = capi.DeviceBVH(gpu_ctx, sc).build()
    assert gb.read_morton_pairs().tobytes() == ob.pairs.tobytes()
    res = (3840, 2160)
    cam = osc.demo_camera(*res)
    want, _ = ob.trace_primary(cam)
    ccam = as_capi_camera(cam)
    rank_of, slot_of = capi.tile_slot_map(res[0], res[1], 8)
    for r in range(8):
        slots = capi.tiles_owned_pixels(ccam, r, 8)
        dc = gpu_ctx.alloc(slots * 32)
        gb.generate_contacts_primary_tiled_d(ccam, r, 8, dc)
        compact = gpu_ctx.download(dc, slots, capi.CONTACT)
        gpu_ctx.free(dc)
        mine = rank_of == r
        assert compact[slot_of[mine]].tobytes() == want[mine].tobytes(), 'rank %d' % r
    gb.close(); sc.close()
