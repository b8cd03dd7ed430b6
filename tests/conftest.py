import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

MODELS = ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The built libraries are git-ignored artefacts: when a checkout lacks them (fresh clone), build them once
    (hipcc cross-compiles gfx950 without a GPU).  The product itself never builds or falls back on its own."""
    need = [os.path.join(ROOT, 'openclraytracer_amd', 'librtmi.so'), os.path.join(ROOT, 'openclraytracer_amd', 'librtmi_host.so'),
            os.path.join(ROOT, 'apps', 'bin', 'rt_demo'), os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool'),
            os.path.join(ROOT, 'oracle', '_build', 'librt_oracle.so')]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def scene_path(name):
    return os.path.join(ROOT, 'assets', 'scenes', name + '.txt')


_scene_cache = {}


def packed_scene(name):
    """Packed scene buffer through the ORACLE's loader (test infrastructure)."""
    import scene_pack
    if name not in _scene_cache:
        cwd = os.getcwd()
        os.chdir(ROOT)
        try:
            _scene_cache[name] = scene_pack.load_scene(scene_path(name))
        finally:
            os.chdir(cwd)
    return _scene_cache[name]


@pytest.fixture(scope='session')
def gpu_ctx():
    from openclraytracer_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


_oracle_bvh = {}


def oracle_bvh(name):
    import oracle as orc
    if name not in _oracle_bvh:
        sc = orc.Scene(packed_scene(name))
        _oracle_bvh[name] = (sc, orc.BVH(sc, 0))
    return _oracle_bvh[name]


_gpu_bvh = {}


def gpu_bvh(ctx, name):
    from openclraytracer_amd import capi
    if name not in _gpu_bvh:
        sc = capi.DeviceScene(ctx, packed_scene(name))
        _gpu_bvh[name] = (sc, capi.DeviceBVH(ctx, sc).build())
    return _gpu_bvh[name]


def as_capi_camera(cam):
    from openclraytracer_amd import capi
    return cam.view(capi.CAMERA)
