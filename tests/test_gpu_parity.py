"""GPU parity tests proper: the HIP path through the C ABI vs the CPU oracle (bit-exact for
integer / index work and for t and normals; shaded RGB within 1e-5 relative)."""
import numpy as np
import pytest

from conftest import MODELS, as_capi_camera, gpu_bvh, oracle_bvh

pytestmark = pytest.mark.gpu


def _bytes_equal(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize('name', MODELS)
def test_bvh_build_bit_exact(gpu_ctx, name):
    """morton pairs after the bitonic network and every 48-byte BVHNode equal the oracle's."""
    _, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    info = gb.info()
    assert info['leaves'] == ob.n and info['morton_items'] == ob.N
    pairs = gb.read_morton_pairs()
    assert _bytes_equal(pairs, ob.pairs), 'sorted (morton, leaf) pairs differ'
    nodes = gb.read_nodes()
    diff = np.nonzero(nodes.view(np.uint8).reshape(-1, 48) != ob.nodes.view(np.uint8).reshape(-1, 48))[0]
    assert diff.size == 0, 'first differing node %d' % diff[0]
    rc, depth = ob.check_hierarchy()
    assert rc == 0 and depth == info['max_depth']


@pytest.mark.parametrize('name,res', [('cube', (64, 48)), ('gardenlight', (64, 48)), ('tenryuu', (160, 120)),
                                      ('man', (160, 120)), ('clumsydragon', (128, 96)), ('cube', (641, 479)),
                                      ('gardenlight', (640, 480))])
def test_trace_primary_bit_exact(gpu_ctx, name, res):
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(*res)
    want, _ = ob.trace_primary(cam)
    got = gb.generate_contacts_primary(as_capi_camera(cam))
    assert (got['pixelIndex'] == np.arange(len(got))).all()
    bad = np.nonzero((got.view(np.uint8).reshape(-1, 32) != want.view(np.uint8).reshape(-1, 32)).any(axis=1))[0]
    assert bad.size == 0, '%d contacts differ, first %d: got %s want %s' % (bad.size, bad[0], got[bad[0]], want[bad[0]])


@pytest.mark.parametrize('name', MODELS)
def test_trace_primary_full_hd_bit_exact(gpu_ctx, name):
    """BASELINE.json configs[1]: 1920x1080 primary rays, every contact compared with the oracle."""
    osc, ob = oracle_bvh(name)
    _, gb = gpu_bvh(gpu_ctx, name)
    cam = osc.demo_camera(1920, 1080)
    want, st = ob.trace_primary(cam)
    got = gb.generate_contacts_primary(as_capi_camera(cam))
    assert got.tobytes() == want.tobytes()
    assert int((got['normalAndDistance'][:, 3] > 0).sum()) == st.hits
