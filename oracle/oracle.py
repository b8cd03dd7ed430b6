"""oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes front-end of oracle/_build/librt_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
See rt_oracle.h for the parity status (pinned on the reference's own OpenCL C kernels, tests/golden/reference_cl.json).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RTMI_ORACLE_LIB') or os.path.join(HERE, '_build', 'librt_oracle.so')      # (RTMI_ORACLE_LIB: the sanitizer build, tools/san)

NODE = np.dtype([('data', '<u4', 4), ('lo', '<f4', 4), ('hi', '<f4', 4)])        # 48 B  BVHData.h:70-74
CONTACT = np.dtype([('pixelIndex', '<u4'), ('materialIndex', '<u4'), ('pad', '<u4', 2),
                    ('normal_t', '<f4', 4)])                                      # 32 B  CLStructs.h:112-118
RAY = np.dtype([('idx', '<u4'), ('pad', '<u4', 3), ('origin', '<f4', 4), ('direction', '<f4', 4)])  # 48 B
PAIR = np.dtype([('key', '<u4'), ('value', '<u4')])
CAMERA = np.dtype([('FOVDistance', '<f4'), ('resX', '<u4'), ('resY', '<u4'), ('supersampling', '<u4'),
                   ('m', '<f4', 12)])                                             # 64 B  CLStructs.h:44-51


GRID_DATA = np.dtype([('res', '<u4', 3), ('step', '<f4', 3), ('leafDensity', '<f4'), ('padTo64', '<f4'),
                      ('lo', '<f4', 4), ('hi', '<f4', 4)])                        # 64 B  TwoLevelGridData.h:52-64
TOP_CELL = np.dtype([('res', '<u4', 3), ('firstLeafIdx', '<u4')])                 # 16 B  TwoLevelGridData.h:41-47
TLG_INFO_FIELDS = ('cells', 'pairs_count', 'pairs_pow2', 'leaf_cells_count', 'leaf_pairs_count', 'leaf_pairs_pow2',
                   'written_leaf_pairs', 'dropped_leaf_pairs', 'nonempty_cells', 'flat_cells')


class F4(C.Structure):
    _fields_ = [('x', C.c_float), ('y', C.c_float), ('z', C.c_float), ('w', C.c_float)]


class TLGStats(C.Structure):
    _fields_ = [('top_visits', C.c_uint64), ('leaf_visits', C.c_uint64), ('tri_tests', C.c_uint64),
                ('hits', C.c_uint64), ('dropped_rays', C.c_uint64), ('sum_t', C.c_double), ('repeat_tests', C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [('inner_visits', C.c_uint64), ('leaf_tests', C.c_uint64), ('hits', C.c_uint64),
                ('max_stack', C.c_uint32), ('pad_', C.c_uint32), ('sum_t', C.c_double)]

    def as_dict(self):
        return dict(inner_visits=self.inner_visits, leaf_tests=self.leaf_tests, hits=self.hits,
                    max_stack=self.max_stack, sum_t=self.sum_t)


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ('rt_oracle.c', 'tlg_oracle.c', 'rt_oracle.h', 'tlg_oracle.h') if
           os.path.exists(os.path.join(HERE, f))]
    if os.environ.get('RTMI_ORACLE_LIB'):
        return LIB_PATH
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(['make', '-s', '-C', HERE])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.orc_sizeof.restype = C.c_size_t
        L.orc_sizeof.argtypes = [C.c_char_p]
        L.orc_demo_camera.argtypes = [vp, u32, u32, vp]
        L.orc_generate_ray.argtypes = [vp, u32, vp]
        L.orc_generate_rays.argtypes = [vp, u32, u32, vp]
        L.orc_scene_triangle_count.restype = C.c_uint64
        L.orc_scene_triangle_count.argtypes = [vp]
        L.orc_scene_triangle.argtypes = [vp, u32, vp, vp, vp]
        L.orc_morton_buffer_items.restype = u32
        L.orc_morton_buffer_items.argtypes = [u32]
        L.orc_bitonic_sort.argtypes = [vp, u32]
        L.orc_build_bvh.argtypes = [vp, i32, vp, vp]
        L.orc_morton_leaves.argtypes = [vp, vp, vp]
        L.orc_radix_tree.argtypes = [vp, vp, u32, i32]
        L.orc_refit.argtypes = [vp, u32, i32]
        L.orc_check_hierarchy.argtypes = [vp, u32, vp]
        L.orc_trace_primary.argtypes = [vp, vp, u32, vp, vp, vp, i32]
        L.orc_trace_primary_range.argtypes = [vp, vp, u32, vp, u32, u32, vp, vp, i32]
        L.orc_trace_primary_visits.argtypes = [vp, vp, u32, vp, vp, vp, vp, i32]
        L.orc_trace_rays.argtypes = [vp, u32, vp, u32, vp, vp, vp, i32]
        L.orc_brute_primary.argtypes = [vp, vp, vp, i32]
        L.orc_shade_frame.argtypes = [vp, vp, u32, vp, vp, i32]
        L.orc_scan_inclusive_u32.argtypes = [vp, vp, u32]
        f32 = C.c_float
        L.orc_generate_secondary_rays.restype = u32
        L.orc_generate_secondary_rays.argtypes = [vp, vp, vp, u32, i32, F4, f32, vp]
        L.orc_pow_third.restype = f32
        L.orc_pow_third.argtypes = [f32]
        L.orc_aabb_triangle_intersect.argtypes = [F4, F4, F4, F4, F4]
        L.orc_tlg_grid_data.argtypes = [vp, f32, f32, vp]
        L.orc_tlg_build.restype = vp
        L.orc_tlg_build.argtypes = [vp, f32, f32]
        L.orc_tlg_free.argtypes = [vp]
        L.orc_tlg_info_get.argtypes = [vp, vp]
        for fn in ('grid', 'top_cells', 'top_pairs', 'cell_ranges', 'leaf_pairs', 'leaf_ranges'):
            getattr(L, 'orc_tlg_' + fn).restype = vp
            getattr(L, 'orc_tlg_' + fn).argtypes = [vp]
        L.orc_tlg_trace_primary.argtypes = [vp, vp, vp, vp, i32]
        L.orc_tlg_trace_rays.argtypes = [vp, vp, u32, vp, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ncpu() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    16-CPU share of its 256 hardware threads to a one-GPU job)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


class Scene:
    """A packed scene buffer (bytes, reference format) held for the C oracle."""

    def __init__(self, buf: bytes):
        self.bytes = bytes(buf)
        self.arr = np.frombuffer(self.bytes, dtype=np.uint8)
        self.n_tris = int(lib().orc_scene_triangle_count(_p(self.arr)))

    def demo_camera(self, resx: int, resy: int) -> np.ndarray:
        cam = np.zeros(1, dtype=CAMERA)
        lib().orc_demo_camera(_p(self.arr), resx, resy, _p(cam))
        return cam

    def triangles(self):
        """(verts[n,3,4] f32, refs[n,3] u32, material[n] u32) through the reference's index walkers."""
        n = self.n_tris
        v = np.zeros((n, 3, 4), dtype=np.float32)
        r = np.zeros((n, 3), dtype=np.uint32)
        m = np.zeros(n, dtype=np.uint32)
        L = lib()
        for t in range(n):
            L.orc_scene_triangle(_p(self.arr), t, C.c_void_p(v.ctypes.data + t * 48),
                                 C.c_void_p(r.ctypes.data + t * 12), C.c_void_p(m.ctypes.data + t * 4))
        return v, r, m


class BVH:
    def __init__(self, scene: Scene, mode: int = 0):
        self.scene = scene
        self.n = scene.n_tris
        self.N = int(lib().orc_morton_buffer_items(self.n))
        self.nodes = np.zeros(2 * self.n - 1, dtype=NODE)
        self.pairs = np.zeros(self.N, dtype=PAIR)
        rc = lib().orc_build_bvh(_p(scene.arr), mode, _p(self.nodes), _p(self.pairs))
        if rc:
            raise RuntimeError('orc_build_bvh failed: %d' % rc)

    def check_hierarchy(self):
        d = C.c_uint32(0)
        rc = lib().orc_check_hierarchy(_p(self.nodes), self.n, C.byref(d))
        return rc, d.value

    def trace_primary(self, cam: np.ndarray, nthreads: int = 0, first: int = 0, count: int | None = None, out: np.ndarray | None = None):
        npx = int(cam['resX'][0]) * int(cam['resY'][0])
        if count is None:
            count = npx - first
        if out is None:
            out = np.zeros(count, dtype=CONTACT)
        st = Stats()
        lib().orc_trace_primary_range(_p(cam), _p(self.nodes), self.n, _p(self.scene.arr), first, count,
                                      _p(out), C.byref(st), nthreads or ncpu())
        return out, st

    def trace_primary_visits(self, cam: np.ndarray, nthreads: int = 0):
        npx = int(cam['resX'][0]) * int(cam['resY'][0])
        out = np.zeros(npx, dtype=CONTACT)
        visits = np.zeros(npx, dtype=np.uint32)
        st = Stats()
        lib().orc_trace_primary_visits(_p(cam), _p(self.nodes), self.n, _p(self.scene.arr), _p(out), _p(visits),
                                       C.byref(st), nthreads or ncpu())
        return out, visits, st

    def trace_rays(self, rays: np.ndarray, nthreads: int = 0):
        out = np.zeros(len(rays), dtype=CONTACT)
        st = Stats()
        lib().orc_trace_rays(_p(rays), len(rays), _p(self.nodes), self.n, _p(self.scene.arr), _p(out),
                             C.byref(st), nthreads or ncpu())
        return out, st


def set_device_arith(on: bool) -> None:
    """dot / cross as ROCm's OpenCL C library defines them (fused), i.e. the arithmetic of the reference's DEVICE branch on
    this GPU, instead of the host branch's.  Only tests that compare with the reference's own kernels switch it on."""
    lib().orc_set_device_arith(1 if on else 0)


def build_stages(scene: Scene, mode: int = 0) -> dict:
    """The products of every stage of BVHManager::initializeFrame + construct (BVHManager.cpp:122-228), so that each can be
    compared with what the reference's own kernel makes of the same input (tests/test_gpu_reference_cl.py):
    leaves / pairs after calculateMortonCodes, sorted pairs, nodes after buildRadixTree, nodes after computeBoundingBoxes."""
    L = lib()
    n = scene.n_tris
    N = int(L.orc_morton_buffer_items(n))
    nodes = np.zeros(2 * n - 1, dtype=NODE)
    pairs = np.full(N, 0xFFFFFFFF, dtype=np.uint32).repeat(2).view(PAIR).copy()
    L.orc_morton_leaves(_p(scene.arr), _p(nodes), _p(pairs))
    out = dict(n=n, N=N, leaves=nodes.copy(), pairs=pairs.copy())
    L.orc_bitonic_sort(_p(pairs), N)
    out['sorted'] = pairs.copy()
    L.orc_radix_tree(_p(nodes), _p(pairs), n, mode)
    out['radix'] = nodes.copy()
    L.orc_refit(_p(nodes), n, mode)
    out['nodes'] = nodes
    return out


def brute_primary(scene: Scene, cam: np.ndarray, nthreads: int = 0) -> np.ndarray:
    npx = int(cam['resX'][0]) * int(cam['resY'][0])
    out = np.zeros(npx, dtype=CONTACT)
    lib().orc_brute_primary(_p(cam), _p(scene.arr), _p(out), nthreads or ncpu())
    return out


def generate_rays(cam: np.ndarray, first: int = 0, count: int | None = None) -> np.ndarray:
    npx = int(cam['resX'][0]) * int(cam['resY'][0])
    if count is None:
        count = npx - first
    rays = np.zeros(count, dtype=RAY)
    lib().orc_generate_rays(_p(cam), first, count, _p(rays))
    return rays


def shade_frame(scene: Scene, cam: np.ndarray, contacts: np.ndarray, nthreads: int = 0, out: np.ndarray | None = None) -> np.ndarray:
    npx = int(cam['resX'][0]) * int(cam['resY'][0])
    rgba = np.zeros((npx, 4), dtype=np.float32) if out is None else out      # the C side clears it (memset(colors, 0))
    lib().orc_shade_frame(_p(cam), _p(contacts), len(contacts), _p(scene.arr), _p(rgba), nthreads or ncpu())
    return rgba


def bitonic_sort(pairs: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(pairs.copy())
    lib().orc_bitonic_sort(_p(out), len(out))
    return out


def scan_inclusive(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    lib().orc_scan_inclusive_u32(_p(a), _p(out), len(a))
    return out


def _view(ptr, dtype, count):
    """copy of `count` records behind a C pointer"""
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (dtype.itemsize * count)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class TLG:
    """Two-Level Grid built by the C oracle (tlg_oracle.c): TwoLevelGridManager::initializeFrame + construct."""

    def __init__(self, scene: Scene, top_density: float = 2.0, leaf_density: float = 2.0):
        self.scene = scene
        self.h = lib().orc_tlg_build(_p(scene.arr), top_density, leaf_density)
        if not self.h:
            raise RuntimeError('orc_tlg_build failed (degenerate scene box)')
        info = np.zeros(len(TLG_INFO_FIELDS), dtype=np.uint32)
        lib().orc_tlg_info_get(self.h, _p(info))
        self.info = dict(zip(TLG_INFO_FIELDS, (int(x) for x in info)))
        L, i = lib(), self.info
        self.grid = _view(L.orc_tlg_grid(self.h), GRID_DATA, 1)
        self.top_cells = _view(L.orc_tlg_top_cells(self.h), TOP_CELL, i['cells'])
        self.top_pairs = _view(L.orc_tlg_top_pairs(self.h), PAIR, i['pairs_pow2'])
        self.cell_ranges = _view(L.orc_tlg_cell_ranges(self.h), PAIR, i['cells'])
        self.leaf_pairs = _view(L.orc_tlg_leaf_pairs(self.h), PAIR, i['leaf_pairs_pow2'])
        self.leaf_ranges = _view(L.orc_tlg_leaf_ranges(self.h), PAIR, i['leaf_cells_count'])

    def __del__(self):
        try:
            if self.h:
                lib().orc_tlg_free(self.h)
                self.h = None
        except Exception:
            pass

    def trace_primary(self, cam: np.ndarray, nthreads: int = 0):
        n = int(cam['resX'][0]) * int(cam['resY'][0])
        out = np.zeros(n, dtype=CONTACT)
        st = TLGStats()
        lib().orc_tlg_trace_primary(self.h, _p(cam), _p(out), C.byref(st), nthreads or ncpu())
        return out, st.as_dict()

    def trace_rays(self, rays: np.ndarray, nthreads: int = 0):
        out = np.zeros(len(rays), dtype=CONTACT)
        st = TLGStats()
        if len(rays):
            lib().orc_tlg_trace_rays(self.h, _p(rays), len(rays), _p(out), C.byref(st), nthreads or ncpu())
        return out, st.as_dict()


def pow_third(x: float) -> float:
    return float(lib().orc_pow_third(C.c_float(x)))


def generate_secondary_rays(contacts: np.ndarray, kind: int, target=(0.0, 0.0, 0.0), eps: float = 1e-3,
                            cam: np.ndarray | None = None, rays_in: np.ndarray | None = None) -> np.ndarray:
    """rays for the next bounce (extension, reference TODO 2.2): kind 0 shadow rays towards `target`, 1 reflections"""
    out = np.zeros(len(contacts), dtype=RAY)
    n = lib().orc_generate_secondary_rays(_p(cam) if cam is not None else None, _p(rays_in) if rays_in is not None else None,
                                          _p(contacts), len(contacts), kind, F4(target[0], target[1], target[2], 0.0), eps, _p(out))
    return out[:n].copy()
