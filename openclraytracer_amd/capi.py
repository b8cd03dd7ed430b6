"""ctypes binding of librtmi.so -- one Python function/class per entry point of include/rtmi.h.

This is harness plumbing for tests/ and bench.py (the reference's host language is C++; the C++
mirror of its classes lives in openclraytracer_amd/csrc/host/).  There is NO CPU fallback: if the
HIP library is missing or no gfx950 device is present, loading / context creation raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, 'librtmi.so')

# byte-identical mirrors of the reference PODs (include/rtmi.h)
CAMERA = np.dtype([('FOVDistance', '<f4'), ('resX', '<u4'), ('resY', '<u4'), ('supersamplingFactor', '<u4'),
                   ('viewTransform', '<f4', 12)])
RAY = np.dtype([('idx', '<u4'), ('pad_', '<u4', 3), ('origin', '<f4', 4), ('direction', '<f4', 4)])
CONTACT = np.dtype([('pixelIndex', '<u4'), ('materialIndex', '<u4'), ('pad', '<u4', 2),
                    ('normalAndDistance', '<f4', 4)])
BVH_NODE = np.dtype([('data', '<u4', 4), ('boundsMin', '<f4', 4), ('boundsMax', '<f4', 4)])
PAIR = np.dtype([('key', '<u4'), ('value', '<u4')])
GRID_DATA = np.dtype([('res', '<u4', 3), ('step', '<f4', 3), ('leafDensity', '<f4'), ('padTo64', '<f4'),
                      ('box_lo', '<f4', 4), ('box_hi', '<f4', 4)])
TOP_CELL = np.dtype([('res', '<u4', 3), ('firstLeafIdx', '<u4')])
TLG_INFO_FIELDS = ('cells', 'pairs_count', 'pairs_pow2', 'leaf_cells_count', 'leaf_pairs_count', 'leaf_pairs_pow2',
                   'written_leaf_pairs')
assert GRID_DATA.itemsize == 64 and TOP_CELL.itemsize == 16
assert CAMERA.itemsize == 64 and RAY.itemsize == 48 and CONTACT.itemsize == 32 and BVH_NODE.itemsize == 48

# every symbol include/rtmi.h declares (tests check the library exports all of them)
SYMBOLS = [
    'rt_abi_version', 'rt_ctx_create', 'rt_ctx_destroy', 'rt_last_error', 'rt_ctx_set_stream', 'rt_ctx_get_stream',
    'rt_ctx_synchronize', 'rt_device_synchronize', 'rt_ctx_device_info', 'rt_buf_alloc', 'rt_buf_free', 'rt_buf_upload', 'rt_buf_download',
    'rt_buf_fill_u32', 'rt_scene_upload', 'rt_scene_upload_indexed', 'rt_scene_destroy', 'rt_scene_device_ptr', 'rt_scene_bytes',
    'rt_scene_triangle_count', 'rt_bvh_create', 'rt_bvh_destroy', 'rt_bvh_initialize_frame', 'rt_bvh_construct', 'rt_bvh_graph_builds',
    'rt_bvh_generate_contacts_primary', 'rt_bvh_generate_contacts_rays', 'rt_bvh_rays_strategy', 'rt_bvh_rays_calibrations', 'rt_bvh_read_nodes',
    'rt_bvh_read_morton_pairs', 'rt_bvh_read_search_nodes', 'rt_bvh_info', 'rt_bvh_trace_strategy', 'rt_bvh_calibration_launches', 'rt_bvh_host_plane_reads', 'rt_bvh_calibration_gpu_ms', 'rt_bvh_block_order', 'rt_bvh_search_tree', 'rt_shade_blinn_phong', 'rt_tiles_owned_pixels',
    'rt_bvh_generate_contacts_primary_tiled', 'rt_shade_blinn_phong_tiled', 'rt_untile_rgba', 'rt_frame_graph_create', 'rt_frame_graph_launch', 'rt_frame_graph_destroy', 'rt_frame_graph_set_camera', 'rt_frame_graph_last_launch', 'rt_frame_graph_state', 'rt_render_frame', 'rt_shade_blinn_phong_frame', 'rt_shade_blinn_phong_tiled_packed', 'rt_untile_rgba_packed',
    'rt_sort_pairs_bitonic', 'rt_sort_pairs', 'rt_scan_inclusive_u32', 'rt_ctx_set_profiling', 'rt_ctx_set_frames_in_flight', 'rt_ctx_trace_kernel_times', 'rt_ctx_last_trace_counters',
    'rt_timer_create', 'rt_timer_start', 'rt_timer_stop',
    'rt_timer_elapsed_ms', 'rt_timer_destroy',
    'rt_tlg_create', 'rt_tlg_destroy', 'rt_tlg_set_densities', 'rt_tlg_initialize_frame', 'rt_tlg_construct',
    'rt_tlg_generate_contacts_primary', 'rt_tlg_generate_contacts_rays', 'rt_tlg_generate_contacts_primary_tiled', 'rt_tlg_generate_frame',
    'rt_comm_unique_id', 'rt_comm_create', 'rt_comm_create_group', 'rt_comm_share', 'rt_comm_destroy', 'rt_comm_rank', 'rt_comm_size', 'rt_comm_uses_rccl', 'rt_comm_rccl_ranks', 'rt_multi_frame_create_rooted', 'rt_multi_frame_root', 'rt_multi_frame_set_camera', 'rt_multi_frame_set_timing', 'rt_multi_frame_times',
    'rt_multi_frame_create', 'rt_multi_frame_render', 'rt_multi_frame_wait', 'rt_multi_frame_pixels', 'rt_multi_frame_layout', 'rt_multi_frame_destroy',
    'rt_tlg_info_get', 'rt_tlg_rebuilds_without_readback', 'rt_tlg_walk_records', 'rt_tlg_read', 'rt_tlg_pow_third', 'rt_generate_secondary_rays', 'rt_generate_primary_rays',
]


class RtError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """Loads librtmi.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get('RTMI_LIBRARY'):          # tests: the twin with tiny packet stacks (csrc/Makefile), same ABI
        LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), os.path.basename(os.environ['RTMI_LIBRARY']))
    if not os.path.exists(LIB_PATH):
        raise RtError('librtmi.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                      'or `make -C openclraytracer_amd/csrc`' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    pvp = C.POINTER(C.c_void_p)
    L.rt_abi_version.restype = i32
    L.rt_ctx_create.argtypes = [i32, pvp]
    L.rt_ctx_destroy.argtypes = [vp]; L.rt_ctx_destroy.restype = None
    L.rt_last_error.argtypes = [vp]; L.rt_last_error.restype = C.c_char_p
    L.rt_ctx_set_stream.argtypes = [vp, vp]
    L.rt_ctx_get_stream.argtypes = [vp]; L.rt_ctx_get_stream.restype = vp
    L.rt_ctx_synchronize.argtypes = [vp]
    L.rt_device_synchronize.argtypes = [i32]
    L.rt_ctx_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64), C.c_char_p, C.c_size_t]
    L.rt_buf_alloc.argtypes = [vp, u64, pvp]
    L.rt_buf_free.argtypes = [vp, vp]
    L.rt_buf_upload.argtypes = [vp, vp, vp, u64]
    L.rt_buf_download.argtypes = [vp, vp, vp, u64]
    L.rt_buf_fill_u32.argtypes = [vp, vp, u32, u64]
    L.rt_scene_upload.argtypes = [vp, vp, u64, pvp]
    L.rt_scene_upload_indexed.argtypes = [vp, vp, u64, i32, pvp]
    L.rt_scene_destroy.argtypes = [vp]; L.rt_scene_destroy.restype = None
    L.rt_scene_device_ptr.argtypes = [vp]; L.rt_scene_device_ptr.restype = vp
    L.rt_scene_bytes.argtypes = [vp]; L.rt_scene_bytes.restype = u64
    L.rt_scene_triangle_count.argtypes = [vp]; L.rt_scene_triangle_count.restype = u32
    L.rt_bvh_create.argtypes = [vp, vp, i32, pvp]
    L.rt_bvh_destroy.argtypes = [vp]; L.rt_bvh_destroy.restype = None
    L.rt_bvh_initialize_frame.argtypes = [vp]
    L.rt_bvh_construct.argtypes = [vp]
    L.rt_bvh_generate_contacts_primary.argtypes = [vp, vp, vp]
    L.rt_bvh_generate_contacts_rays.argtypes = [vp, vp, u32, vp]
    L.rt_bvh_read_nodes.argtypes = [vp, vp]
    L.rt_bvh_read_morton_pairs.argtypes = [vp, vp, C.POINTER(u32)]
    L.rt_bvh_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.rt_bvh_trace_strategy.argtypes = [vp, C.POINTER(i32), C.POINTER(u32)]
    L.rt_bvh_search_tree.argtypes = [vp, C.POINTER(u32), C.POINTER(i32)]
    L.rt_shade_blinn_phong.argtypes = [vp, vp, vp, vp, u32, vp]
    L.rt_tiles_owned_pixels.argtypes = [vp, u32, u32]; L.rt_tiles_owned_pixels.restype = u32
    L.rt_bvh_generate_contacts_primary_tiled.argtypes = [vp, vp, u32, u32, vp]
    L.rt_shade_blinn_phong_tiled.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    L.rt_untile_rgba.argtypes = [vp, vp, u32, vp, u32, vp]
    L.rt_shade_blinn_phong_frame.argtypes = [vp, vp, vp, vp, vp]
    L.rt_frame_graph_create.argtypes = [vp, vp, vp, u32, u32, i32, vp, vp, u32, pvp]
    L.rt_frame_graph_launch.argtypes = [vp]
    L.rt_render_frame.argtypes = [vp, vp, vp, u32, u32, i32, vp, vp, u32]
    L.rt_frame_graph_set_camera.argtypes = [vp, vp]
    L.rt_frame_graph_last_launch.argtypes = [vp]
    L.rt_frame_graph_destroy.argtypes = [vp]; L.rt_frame_graph_destroy.restype = None
    L.rt_shade_blinn_phong_tiled_packed.argtypes = [vp, vp, vp, u32, u32, vp, vp, u32]
    L.rt_untile_rgba_packed.argtypes = [vp, vp, u32, vp, u32, u64, vp]
    L.rt_sort_pairs_bitonic.argtypes = [vp, vp, u32]
    L.rt_scan_inclusive_u32.argtypes = [vp, vp, vp, u32]
    L.rt_sort_pairs.argtypes = [vp, vp, u32, i32]
    L.rt_ctx_set_profiling.argtypes = [vp, i32]
    L.rt_ctx_set_frames_in_flight.argtypes = [vp, u32]
    L.rt_ctx_trace_kernel_times.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rt_ctx_last_trace_counters.argtypes = [vp, C.POINTER(C.c_uint32), u32]
    L.rt_timer_create.argtypes = [vp, pvp]
    L.rt_timer_start.argtypes = [vp, vp]
    L.rt_timer_stop.argtypes = [vp, vp]
    L.rt_timer_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.rt_timer_destroy.argtypes = [vp, vp]; L.rt_timer_destroy.restype = None
    L.rt_tlg_create.argtypes = [vp, vp, pvp]
    L.rt_tlg_destroy.argtypes = [vp]; L.rt_tlg_destroy.restype = None
    L.rt_tlg_set_densities.argtypes = [vp, C.c_float, C.c_float]
    L.rt_tlg_initialize_frame.argtypes = [vp]
    L.rt_tlg_construct.argtypes = [vp]
    L.rt_tlg_generate_contacts_primary.argtypes = [vp, vp, vp]
    L.rt_tlg_generate_frame.argtypes = [vp, vp, vp, vp]
    L.rt_tlg_generate_contacts_rays.argtypes = [vp, vp, u32, vp]
    L.rt_tlg_generate_contacts_primary_tiled.argtypes = [vp, vp, u32, u32, vp]
    L.rt_tlg_info_get.argtypes = [vp, vp]
    L.rt_comm_unique_id.argtypes = [vp]
    L.rt_comm_create.argtypes = [vp, vp, i32, i32, pvp]
    L.rt_comm_create_group.argtypes = [pvp, i32, pvp]
    L.rt_comm_share.argtypes = [vp, vp, pvp]
    L.rt_comm_destroy.argtypes = [vp]; L.rt_comm_destroy.restype = None
    L.rt_comm_rank.argtypes = [vp]; L.rt_comm_size.argtypes = [vp]; L.rt_comm_uses_rccl.argtypes = [vp]
    L.rt_multi_frame_create.argtypes = [vp, vp, vp, vp, u32, pvp]
    L.rt_multi_frame_render.argtypes = [pvp, i32]
    L.rt_multi_frame_wait.argtypes = [vp]
    L.rt_multi_frame_pixels.argtypes = [vp]; L.rt_multi_frame_pixels.restype = vp
    L.rt_multi_frame_layout.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.rt_multi_frame_destroy.argtypes = [vp]; L.rt_multi_frame_destroy.restype = None
    L.rt_tlg_rebuilds_without_readback.argtypes = [vp]; L.rt_tlg_rebuilds_without_readback.restype = u32
    L.rt_tlg_read.argtypes = [vp, i32, vp, u64]
    L.rt_tlg_pow_third.argtypes = [C.c_float]; L.rt_tlg_pow_third.restype = C.c_float
    L.rt_generate_secondary_rays.argtypes = [vp, vp, vp, vp, u32, i32, C.POINTER(C.c_float), C.c_float, vp, C.POINTER(u32)]
    L.rt_generate_primary_rays.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _p(a: np.ndarray) -> C.c_void_p:
    return a.ctypes.data_as(C.c_void_p)


def device_synchronize(device: int = 0):
    """every stream of the device (hipDeviceSynchronize)"""
    L = load_library()
    if L.rt_device_synchronize(device):
        raise RtError(L.rt_last_error(None).decode())


class Context:
    """rt_ctx: stands where the reference has OpenCLUtils::CLExecutionContext."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        if self.lib.rt_ctx_create(device, C.byref(h)):
            raise RtError(self.lib.rt_last_error(None).decode())
        self.h = h
        if stream is not None:
            self.check(self.lib.rt_ctx_set_stream(self.h, C.c_void_p(stream)))

    def check(self, rc: int):
        if rc:
            raise RtError(self.lib.rt_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.rt_ctx_destroy(self.h)
            self.h = None

    def synchronize(self):
        self.check(self.lib.rt_ctx_synchronize(self.h))

    def device_info(self) -> dict:
        cu, ws, lds = C.c_int(), C.c_int(), C.c_uint64()
        name = C.create_string_buffer(64)
        self.check(self.lib.rt_ctx_device_info(self.h, C.byref(cu), C.byref(ws), C.byref(lds), name, 64))
        return dict(compute_units=cu.value, wave_size=ws.value, lds_bytes=lds.value, name=name.value.decode())

    def set_frames_in_flight(self, n):
        """1 (default): a frame has the device to itself -- the four-wide kernel starts a frame's most expensive blocks first, sharing
        strategies compete at face value; > 1: several frames in flight -- index order, sharing handicapped (rtmi.h,
        rt_ctx_set_frames_in_flight)."""
        self.check(self.lib.rt_ctx_set_frames_in_flight(self.h, int(n)))

    def set_profiling(self, on):
        """False/0 off, True/1 HIP events around the traversal kernels, 2 the same with the '<.., 2>' kernel names."""
        self.check(self.lib.rt_ctx_set_profiling(self.h, 2 if on == 2 else (1 if on else 0)))

    def last_trace_counters(self) -> dict:
        out = (C.c_uint32 * 8)()
        self.check(self.lib.rt_ctx_last_trace_counters(self.h, out, 8))
        return dict(zip(('deferred_rays', 'split_rays', 'spill_areas_taken', 'wave_per_ray_steps', 'wave_per_ray_steps_max', 'wave_per_ray_nodes',
                         'shared_blocks', 'shared_tasks'), list(out)))

    def trace_kernel_times(self) -> dict:
        n, a, b = C.c_uint64(), C.c_double(), C.c_double()
        self.check(self.lib.rt_ctx_trace_kernel_times(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return dict(launches=n.value, lane_phase_ms=a.value, wave_phase_ms=b.value)

    def get_stream(self) -> int:
        self.lib.rt_ctx_get_stream.restype = C.c_void_p
        return int(self.lib.rt_ctx_get_stream(self.h) or 0)

    def set_stream(self, stream: int | None):
        self.check(self.lib.rt_ctx_set_stream(self.h, C.c_void_p(stream or 0)))

    # ---- CLBuffer-like helpers
    def alloc(self, nbytes: int) -> int:
        d = C.c_void_p()
        self.check(self.lib.rt_buf_alloc(self.h, nbytes, C.byref(d)))
        return d.value

    def free(self, dptr: int):
        self.check(self.lib.rt_buf_free(self.h, C.c_void_p(dptr)))

    def upload(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.rt_buf_upload(self.h, C.c_void_p(dptr), _p(arr), arr.nbytes))

    def download(self, dptr: int, count: int, dtype) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        self.check(self.lib.rt_buf_download(self.h, _p(out), C.c_void_p(dptr), out.nbytes))
        return out

    def fill_u32(self, dptr: int, pattern: int, count: int):
        self.check(self.lib.rt_buf_fill_u32(self.h, C.c_void_p(dptr), pattern, count))

    # ---- primitives
    def sort_pairs_bitonic(self, pairs: np.ndarray) -> np.ndarray:
        d = self.alloc(pairs.nbytes)
        try:
            self.upload(d, pairs)
            self.check(self.lib.rt_sort_pairs_bitonic(self.h, C.c_void_p(d), len(pairs)))
            return self.download(d, len(pairs), PAIR)
        finally:
            self.free(d)

    def sort_pairs(self, pairs: np.ndarray, algorithm: int = 1) -> np.ndarray:
        """rt_sort_pairs: 0 the reference's bitonic network (power-of-two counts), 1 the stable radix sort (any count)"""
        d = self.alloc(max(pairs.nbytes, 8))
        try:
            self.upload(d, pairs)
            self.check(self.lib.rt_sort_pairs(self.h, C.c_void_p(d), len(pairs), algorithm))
            return self.download(d, len(pairs), PAIR)
        finally:
            self.free(d)

    def scan_inclusive(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        di, do = self.alloc(max(a.nbytes, 4)), self.alloc(max(a.nbytes, 4))
        try:
            self.upload(di, a)
            self.check(self.lib.rt_scan_inclusive_u32(self.h, C.c_void_p(di), C.c_void_p(do), len(a)))
            return self.download(do, len(a), np.uint32)
        finally:
            self.free(di); self.free(do)


class Timer:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.lib.rt_timer_create(ctx.h, C.byref(h)))
        self.h = h

    def start(self):
        self.ctx.check(self.ctx.lib.rt_timer_start(self.ctx.h, self.h))

    def stop(self):
        self.ctx.check(self.ctx.lib.rt_timer_stop(self.ctx.h, self.h))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        self.ctx.check(self.ctx.lib.rt_timer_elapsed_ms(self.ctx.h, self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.ctx.lib.rt_timer_destroy(self.ctx.h, self.h)
            self.h = None


class DeviceScene:
    """rt_scene: Scene::loadToGPU of a packed scene buffer."""

    def __init__(self, ctx: Context, scene_bytes: bytes, index_bytes: int = 2):
        self.ctx = ctx
        self.host = np.frombuffer(bytes(scene_bytes), dtype=np.uint8)
        h = C.c_void_p()
        if index_bytes == 2:
            ctx.check(ctx.lib.rt_scene_upload(ctx.h, _p(self.host), self.host.nbytes, C.byref(h)))
        else:       # 4: the 32-bit index extension (meshes with more than 65 535 unique vertices)
            ctx.check(ctx.lib.rt_scene_upload_indexed(ctx.h, _p(self.host), self.host.nbytes, index_bytes, C.byref(h)))
        self.h = h
        self.n_tris = ctx.lib.rt_scene_triangle_count(h)

    def close(self):
        if self.h:
            self.ctx.lib.rt_scene_destroy(self.h)
            self.h = None


class DeviceBVH:
    """rt_bvh: the device side of BVHManager (create = ctor+initialize)."""

    def __init__(self, ctx: Context, scene: DeviceScene, mode: int = 0):
        self.ctx, self.scene = ctx, scene
        h = C.c_void_p()
        ctx.check(ctx.lib.rt_bvh_create(ctx.h, scene.h, mode, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.rt_bvh_destroy(self.h)
            self.h = None

    def initialize_frame(self):
        self.ctx.check(self.ctx.lib.rt_bvh_initialize_frame(self.h))

    def construct(self):
        self.ctx.check(self.ctx.lib.rt_bvh_construct(self.h))

    def build(self):
        self.initialize_frame()
        self.construct()
        return self

    def graph_builds(self) -> int:
        self.ctx.lib.rt_bvh_graph_builds.argtypes = [C.c_void_p]
        self.ctx.lib.rt_bvh_graph_builds.restype = C.c_uint32
        return int(self.ctx.lib.rt_bvh_graph_builds(self.h))

    def info(self) -> dict:
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.ctx.check(self.ctx.lib.rt_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(leaves=a.value, morton_items=b.value, max_depth=c.value)

    def trace_strategy(self) -> dict:
        k, v = C.c_int(), C.c_uint32()
        self.ctx.check(self.ctx.lib.rt_bvh_trace_strategy(self.h, C.byref(k), C.byref(v)))
        return dict(kernel={0: 'lane-per-ray', 1: 'packet', 2: 'packet4'}[k.value], budget=v.value)

    def rays_strategy(self) -> dict:
        """rt_bvh_rays_strategy: how ray buffers are traced (calibrated on the first large buffer)"""
        k, v, n, ms = C.c_int(), C.c_uint32(), C.c_uint32(), C.c_float()
        self.ctx.lib.rt_bvh_rays_strategy.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        self.ctx.check(self.ctx.lib.rt_bvh_rays_strategy(self.h, C.byref(k), C.byref(v), C.byref(n), C.byref(ms)))
        return dict(kernel={0: 'lane-per-ray', 2: 'packet4'}[k.value], budget=v.value, calibration_launches=n.value, calibration_gpu_ms=round(ms.value, 4))

    def rays_calibrations(self) -> dict:
        """rt_bvh_rays_calibrations: calibrations of the ray-buffer entry held now, and how many were retired as stale"""
        n, r = C.c_uint32(), C.c_uint32()
        self.ctx.lib.rt_bvh_rays_calibrations.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.ctx.check(self.ctx.lib.rt_bvh_rays_calibrations(self.h, C.byref(n), C.byref(r)))
        return dict(entries=n.value, recalibrations=r.value)

    def calibration_launches(self) -> int:
        self.ctx.lib.rt_bvh_calibration_launches.argtypes = [C.c_void_p]
        self.ctx.lib.rt_bvh_calibration_launches.restype = C.c_uint32
        return int(self.ctx.lib.rt_bvh_calibration_launches(self.h))

    def block_order(self, capacity=1 << 18):
        """(order, cost) of the eager four-wide frames' blocks, or None (rtmi.h, rt_bvh_block_order)"""
        order, cost, n = np.empty(capacity, np.uint32), np.empty(capacity, np.uint32), C.c_uint32()
        self.ctx.lib.rt_bvh_block_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        self.ctx.check(self.ctx.lib.rt_bvh_block_order(self.h, order.ctypes.data, cost.ctypes.data, capacity, C.byref(n)))
        return (order[:n.value].copy(), cost[:n.value].copy()) if n.value else None

    def host_plane_reads(self) -> int:
        self.ctx.lib.rt_bvh_host_plane_reads.argtypes = [C.c_void_p]
        self.ctx.lib.rt_bvh_host_plane_reads.restype = C.c_uint32
        return int(self.ctx.lib.rt_bvh_host_plane_reads(self.h))

    def calibration_gpu_ms(self) -> float:
        self.ctx.lib.rt_bvh_calibration_gpu_ms.argtypes = [C.c_void_p]
        self.ctx.lib.rt_bvh_calibration_gpu_ms.restype = C.c_float
        return float(self.ctx.lib.rt_bvh_calibration_gpu_ms(self.h))

    def search_tree(self) -> dict:
        d, u = C.c_uint32(), C.c_int()
        self.ctx.check(self.ctx.lib.rt_bvh_search_tree(self.h, C.byref(d), C.byref(u)))
        return dict(depth=d.value, used_by_last_camera=u.value)   # 0 no, 1 every ray, 2 all but the rays with a zero direction component on a plane axis

    def trace_budget(self) -> int:
        return self.trace_strategy()['budget']

    def read_nodes(self) -> np.ndarray:
        n = self.info()['leaves']
        out = np.zeros(2 * n - 1, dtype=BVH_NODE)
        self.ctx.check(self.ctx.lib.rt_bvh_read_nodes(self.h, _p(out)))
        return out

    def read_search_nodes(self) -> np.ndarray:
        n = self.info()['leaves']
        out = np.zeros(2 * n - 1, dtype=BVH_NODE)
        self.ctx.lib.rt_bvh_read_search_nodes.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx.check(self.ctx.lib.rt_bvh_read_search_nodes(self.h, _p(out)))
        return out

    def read_morton_pairs(self) -> np.ndarray:
        N = self.info()['morton_items']
        out = np.zeros(N, dtype=PAIR)
        self.ctx.check(self.ctx.lib.rt_bvh_read_morton_pairs(self.h, _p(out), None))
        return out

    # ---- device-pointer entry points (used by bench.py with torch tensors)
    def generate_contacts_primary_d(self, cam: np.ndarray, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_bvh_generate_contacts_primary(self.h, _p(cam), C.c_void_p(d_contacts)))

    def generate_contacts_primary_tiled_d(self, cam: np.ndarray, rank: int, nranks: int, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_bvh_generate_contacts_primary_tiled(self.h, _p(cam), rank, nranks,
                                                                           C.c_void_p(d_contacts)))

    def generate_contacts_rays_d(self, d_rays: int, count: int, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_bvh_generate_contacts_rays(self.h, C.c_void_p(d_rays), count,
                                                                  C.c_void_p(d_contacts)))

    # ---- convenience host round trips (tests)
    def generate_contacts_primary(self, cam: np.ndarray) -> np.ndarray:
        npx = int(cam['resX'][0]) * int(cam['resY'][0])
        d = self.ctx.alloc(npx * 32)
        try:
            self.ctx.fill_u32(d, 0xDEADBEEF, npx * 8)
            self.generate_contacts_primary_d(cam, d)
            return self.ctx.download(d, npx, CONTACT)
        finally:
            self.ctx.free(d)

    def generate_contacts_rays(self, rays: np.ndarray) -> np.ndarray:
        dr, dc = self.ctx.alloc(max(rays.nbytes, 48)), self.ctx.alloc(max(len(rays) * 32, 32))
        try:
            self.ctx.upload(dr, rays)
            self.ctx.fill_u32(dc, 0xDEADBEEF, max(len(rays) * 8, 8))
            self.generate_contacts_rays_d(dr, len(rays), dc)
            return self.ctx.download(dc, len(rays), CONTACT)
        finally:
            self.ctx.free(dr); self.ctx.free(dc)


class DeviceTLG:
    """rt_tlg: the device side of TwoLevelGridManager (create = ctor+initialize)."""
    GRID_DATA, TOP_CELLS, TOP_PAIRS, CELL_RANGES, LEAF_PAIRS, LEAF_RANGES = range(6)

    def __init__(self, ctx: Context, scene: DeviceScene, top_density: float = 2.0, leaf_density: float = 2.0):
        self.ctx, self.scene = ctx, scene
        h = C.c_void_p()
        ctx.check(ctx.lib.rt_tlg_create(ctx.h, scene.h, C.byref(h)))
        self.h = h
        ctx.check(ctx.lib.rt_tlg_set_densities(self.h, top_density, leaf_density))

    def close(self):
        if self.h:
            self.ctx.lib.rt_tlg_destroy(self.h)
            self.h = None

    def initialize_frame(self):
        self.ctx.check(self.ctx.lib.rt_tlg_initialize_frame(self.h))

    def construct(self):
        self.ctx.check(self.ctx.lib.rt_tlg_construct(self.h))

    def build(self):
        self.initialize_frame()
        self.construct()
        return self

    def info(self) -> dict:
        a = np.zeros(len(TLG_INFO_FIELDS), dtype=np.uint32)
        self.ctx.check(self.ctx.lib.rt_tlg_info_get(self.h, _p(a)))
        return dict(zip(TLG_INFO_FIELDS, (int(x) for x in a)))

    def set_densities(self, top_density: float, leaf_density: float):
        self.ctx.check(self.ctx.lib.rt_tlg_set_densities(self.h, top_density, leaf_density))

    def walk_records(self) -> bool:
        """rt_tlg_walk_records: this build is walked with the walk's own cell / plane records (short division sequences)"""
        self.ctx.lib.rt_tlg_walk_records.argtypes = [C.c_void_p]
        return bool(self.ctx.lib.rt_tlg_walk_records(self.h))

    def rebuilds_without_readback(self) -> int:
        return int(self.ctx.lib.rt_tlg_rebuilds_without_readback(self.h))

    def read(self, what: int) -> np.ndarray:
        i = self.info()
        dtype, count = {0: (GRID_DATA, 1), 1: (TOP_CELL, i['cells']), 2: (PAIR, i['pairs_pow2']), 3: (PAIR, i['cells']),
                        4: (PAIR, i['leaf_pairs_pow2']), 5: (PAIR, i['leaf_cells_count'])}[what]
        out = np.zeros(count, dtype=dtype)
        self.ctx.check(self.ctx.lib.rt_tlg_read(self.h, what, _p(out) if count else None, out.nbytes))
        return out

    def generate_contacts_primary_d(self, cam: np.ndarray, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_tlg_generate_contacts_primary(self.h, _p(cam), C.c_void_p(d_contacts)))

    def generate_frame_d(self, cam: np.ndarray, d_contacts: int, d_rgba: int):
        """rt_tlg_generate_frame: contacts and colours of one frame in one launch"""
        self.ctx.check(self.ctx.lib.rt_tlg_generate_frame(self.h, _p(cam), C.c_void_p(d_contacts), C.c_void_p(d_rgba)))

    def generate_contacts_primary_tiled_d(self, cam: np.ndarray, rank: int, nranks: int, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_tlg_generate_contacts_primary_tiled(self.h, _p(cam), rank, nranks,
                                                                           C.c_void_p(d_contacts)))

    def generate_contacts_rays_d(self, d_rays: int, count: int, d_contacts: int):
        self.ctx.check(self.ctx.lib.rt_tlg_generate_contacts_rays(self.h, C.c_void_p(d_rays), count,
                                                                  C.c_void_p(d_contacts)))

    generate_contacts_primary = DeviceBVH.generate_contacts_primary
    generate_contacts_rays = DeviceBVH.generate_contacts_rays


def tlg_pow_third(x: float) -> float:
    return float(load_library().rt_tlg_pow_third(C.c_float(x)))


def shade_blinn_phong_d(ctx: Context, scene: DeviceScene, cam: np.ndarray, d_contacts: int, count: int, d_rgba: int):
    ctx.check(ctx.lib.rt_shade_blinn_phong(ctx.h, scene.h, _p(cam), C.c_void_p(d_contacts), count, C.c_void_p(d_rgba)))


def shade_blinn_phong(ctx: Context, scene: DeviceScene, cam: np.ndarray, contacts: np.ndarray) -> np.ndarray:
    npx = int(cam['resX'][0]) * int(cam['resY'][0])
    dc, dr = ctx.alloc(max(contacts.nbytes, 32)), ctx.alloc(npx * 16)
    try:
        ctx.upload(dc, contacts)
        ctx.fill_u32(dr, 0x7FC00000, npx * 4)
        shade_blinn_phong_d(ctx, scene, cam, dc, len(contacts), dr)
        return ctx.download(dr, npx * 4, np.float32).reshape(npx, 4)
    finally:
        ctx.free(dc); ctx.free(dr)


def generate_secondary_rays(ctx: Context, contacts: np.ndarray, kind: int, target=(0.0, 0.0, 0.0), eps: float = 1e-3,
                            cam: np.ndarray | None = None, rays_in: np.ndarray | None = None) -> np.ndarray:
    """host round trip around rt_generate_secondary_rays (tests)"""
    n = len(contacts)
    dc, dr = ctx.alloc(max(contacts.nbytes, 32)), ctx.alloc(max(n, 1) * 48)
    di = ctx.alloc(max(rays_in.nbytes, 48)) if rays_in is not None else None
    try:
        ctx.upload(dc, contacts)
        if rays_in is not None:
            ctx.upload(di, rays_in)
        ctx.fill_u32(dr, 0xDEADBEEF, max(n, 1) * 12)
        tgt = (C.c_float * 3)(*[float(x) for x in target])
        cnt = C.c_uint32(0)
        ctx.check(ctx.lib.rt_generate_secondary_rays(ctx.h, _p(cam) if cam is not None else None, C.c_void_p(di) if di else None,
                                                     C.c_void_p(dc), n, kind, tgt, eps, C.c_void_p(dr), C.byref(cnt)))
        return ctx.download(dr, cnt.value, RAY)
    finally:
        ctx.free(dc); ctx.free(dr)
        if di:
            ctx.free(di)


def generate_primary_rays_d(ctx: Context, cam: np.ndarray, d_rays: int):
    """rt_generate_primary_rays: resX*resY rays of `cam` into a device buffer of 48 bytes per ray"""
    ctx.check(ctx.lib.rt_generate_primary_rays(ctx.h, _p(cam), C.c_void_p(d_rays)))


def generate_secondary_rays_d(ctx: Context, d_contacts: int, count: int, kind: int, d_rays_out: int, target=(0.0, 0.0, 0.0),
                              eps: float = 1e-3, cam: np.ndarray | None = None, d_rays_in: int | None = None) -> int:
    """rt_generate_secondary_rays on device buffers; -> number of rays written (synchronises)"""
    tgt = (C.c_float * 3)(*[float(x) for x in target])
    cnt = C.c_uint32(0)
    ctx.check(ctx.lib.rt_generate_secondary_rays(ctx.h, _p(cam) if cam is not None else None, C.c_void_p(d_rays_in) if d_rays_in else None,
                                                 C.c_void_p(d_contacts), count, kind, tgt, eps, C.c_void_p(d_rays_out), C.byref(cnt)))
    return int(cnt.value)


def tiles_owned_pixels(cam: np.ndarray, rank: int, nranks: int) -> int:
    return int(load_library().rt_tiles_owned_pixels(_p(cam), rank, nranks))


def shade_blinn_phong_tiled_d(ctx, scene, cam, rank, nranks, d_contacts, d_rgba):
    ctx.check(ctx.lib.rt_shade_blinn_phong_tiled(ctx.h, scene.h, _p(cam), rank, nranks, C.c_void_p(d_contacts),
                                                 C.c_void_p(d_rgba)))


FRAME_UNTILED, FRAME_TILED, FRAME_TILED_PACKED = 0, 1, 2


class FrameGraph:
    """rt_frame_graph: trace + shade of one steady-state frame as a HIP graph (one launch call per replay)."""

    def __init__(self, bvh, scene, cam, d_contacts: int, d_out: int, layout: int = FRAME_UNTILED, rank: int = 0, nranks: int = 1,
                 slots_per_rank: int = 0):
        self.ctx = bvh.ctx
        self.cam = np.array(cam, copy=True)              # the C side reads it only during create()
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.rt_frame_graph_create(bvh.h, scene.h, _p(self.cam), rank, nranks, layout, C.c_void_p(d_contacts),
                                                          C.c_void_p(d_out), slots_per_rank, C.byref(h)))
        self.h = h
        self._launch = self.ctx.lib.rt_frame_graph_launch
        self._set_camera = self.ctx.lib.rt_frame_graph_set_camera

    def launch(self):
        if self._launch(self.h):
            self.ctx.check(1)

    def set_camera(self, cam):
        """rt_frame_graph_set_camera: the next launch renders this camera (same resolution); nothing is captured again"""
        if self._set_camera(self.h, _p(cam)):
            self.ctx.check(1)

    def last_launch(self) -> int:
        """0 steady-state replay, 1 replay behind the per-camera preparation, 2 eager sequence, -1 nothing launched yet"""
        return int(self.ctx.lib.rt_frame_graph_last_launch(self.h))

    def state(self) -> dict:
        """rt_frame_graph_state: the graph's policy state (include/rtmi.h)"""
        out = (C.c_uint32 * 8)()
        self.ctx.lib.rt_frame_graph_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        self.ctx.check(self.ctx.lib.rt_frame_graph_state(self.h, out, 8))
        return dict(zip(('order_interval', 'order_age', 'order_sorted', 'direct', 'eager', 'moved', 'four_wide', 'no_unsafe'), [int(x) for x in out]))

    def close(self):
        if self.h:
            self.ctx.lib.rt_frame_graph_destroy(self.h)
            self.h = None


def render_frame_d(bvh, scene, cam, d_contacts, d_out, rank=0, nranks=1, layout=0, slots_per_rank=0):
    """rt_render_frame: one eager frame, contacts + colours, the trace kernels shading what they store where the strategy allows it"""
    bvh.ctx.check(bvh.ctx.lib.rt_render_frame(bvh.h, scene.h, _p(cam), rank, nranks, layout, C.c_void_p(d_contacts), C.c_void_p(d_out), slots_per_rank))


def shade_blinn_phong_frame_d(ctx, scene, cam, d_primary_contacts, d_rgba):
    ctx.check(ctx.lib.rt_shade_blinn_phong_frame(ctx.h, scene.h, _p(cam), C.c_void_p(d_primary_contacts), C.c_void_p(d_rgba)))


def shade_blinn_phong_tiled_packed_d(ctx, scene, cam, rank, nranks, d_contacts, d_planes, slots_per_rank):
    ctx.check(ctx.lib.rt_shade_blinn_phong_tiled_packed(ctx.h, scene.h, _p(cam), rank, nranks, C.c_void_p(d_contacts),
                                                        C.c_void_p(d_planes), slots_per_rank))


def untile_rgba_packed_d(ctx, cam, nranks, d_gathered, slots_per_rank, d_frame, rank_stride_floats=0):
    ctx.check(ctx.lib.rt_untile_rgba_packed(ctx.h, _p(cam), nranks, C.c_void_p(d_gathered), slots_per_rank,
                                            rank_stride_floats, C.c_void_p(d_frame)))


def untile_rgba_d(ctx, cam, nranks, d_gathered, slots_per_rank, d_frame):
    ctx.check(ctx.lib.rt_untile_rgba(ctx.h, _p(cam), nranks, C.c_void_p(d_gathered), slots_per_rank,
                                     C.c_void_p(d_frame)))


def tiling(nranks: int, root_extra: int = 0) -> int:
    """RT_TILING of include/rtmi.h: the `nranks` word of the tiled entry points"""
    return int(nranks) | (int(root_extra) << 16)


def tile_slot_map(resx: int, resy: int, nranks: int):
    """Host statement of the tiling contract of include/rtmi.h: for every pixel, (rank, slot).  `nranks` is the word of
    the C ABI (low 16 bits N, high 16 bits e = extra tiles per period of N + e tiles for rank 0)."""
    T = 64
    n, e = nranks & 0xFFFF, nranks >> 16
    tiles_x = (resx + T - 1) // T
    y, x = np.divmod(np.arange(resx * resy, dtype=np.int64), resx)
    tx, ty = x // T, y // T
    tile = ty * tiles_x + tx
    q, p = np.divmod(tile, n + e)
    rank = np.where(p <= e, 0, p - e)
    ordinal = np.where(p <= e, q * (e + 1) + p, q)
    slot = ordinal * (T * T) + (y - ty * T) * T + (x - tx * T)
    return rank.astype(np.int64), slot.astype(np.int64)


# ---- the native multi-GPU frame (rt_comm_* / rt_multi_frame_*) ------------------------------------------------------------
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """rank 0 of a job with one process per GPU: the id every rank passes to Comm.create (ncclGetUniqueId)"""
    L = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if L.rt_comm_unique_id(buf):
        raise RtError(L.rt_last_error(None).decode())
    return buf.raw


class Comm:
    """rt_comm: one communicator rank on one context"""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def create(cls, ctx: Context, comm_id: bytes | None, rank: int, nranks: int) -> 'Comm':
        h = C.c_void_p()
        idb = C.create_string_buffer(comm_id, COMM_ID_BYTES) if comm_id is not None else None
        ctx.check(ctx.lib.rt_comm_create(ctx.h, idb, rank, nranks, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def create_group(cls, ctxs) -> list:
        """one process, len(ctxs) contexts (RCCL when their devices differ, device copies when they share one)"""
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        rc = ctxs[0].lib.rt_comm_create_group(arr, n, out)
        if rc:
            raise RtError((ctxs[0].lib.rt_last_error(ctxs[0].h) or ctxs[0].lib.rt_last_error(None) or b'rt_comm_create_group failed').decode())
        return [cls(ctxs[r], C.c_void_p(out[r])) for r in range(n)]

    def share(self, ctx: Context) -> 'Comm':
        """another context of this process and device on the same RCCL communicator (several frames in flight per rank)"""
        h = C.c_void_p()
        ctx.check(ctx.lib.rt_comm_share(self.h, ctx.h, C.byref(h)))
        return Comm(ctx, h)

    def rank(self) -> int:
        return int(self.ctx.lib.rt_comm_rank(self.h))

    def size(self) -> int:
        return int(self.ctx.lib.rt_comm_size(self.h))

    def uses_rccl(self) -> bool:
        return bool(self.ctx.lib.rt_comm_uses_rccl(self.h))

    def rccl_ranks(self) -> int:
        """ncclCommCount of the communicator (0 without RCCL)"""
        self.ctx.lib.rt_comm_rccl_ranks.argtypes = [C.c_void_p]
        return int(self.ctx.lib.rt_comm_rccl_ranks(self.h))

    def close(self):
        if self.h:
            self.ctx.lib.rt_comm_destroy(self.h)
            self.h = None


class MultiFrame:
    """rt_multi_frame: this rank's part of the tiled frame + the gather + (rank 0) the untile"""

    def __init__(self, comm: Comm, bvh, scene, cam: np.ndarray, root_extra: int = 1, root: int = 0):
        self.comm, self.ctx = comm, comm.ctx
        self.cam = np.array(cam, copy=True)
        h = C.c_void_p()
        L = self.ctx.lib
        L.rt_multi_frame_create_rooted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        self.ctx.check(L.rt_multi_frame_create_rooted(comm.h, bvh.h, scene.h, _p(self.cam), root_extra, root, C.byref(h)))
        self.h = h
        self.root = root

    def set_camera(self, cam: np.ndarray):
        """rt_multi_frame_set_camera: the next frames render this camera (every rank passes the same one)"""
        self.ctx.lib.rt_multi_frame_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        self.cam = np.array(cam, copy=True)
        self.ctx.check(self.ctx.lib.rt_multi_frame_set_camera(self.h, _p(self.cam)))

    def set_timing(self, on: bool = True):
        self.ctx.lib.rt_multi_frame_set_timing.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(self.ctx.lib.rt_multi_frame_set_timing(self.h, 1 if on else 0))

    def times(self) -> dict:
        a, g, n = C.c_double(), C.c_double(), C.c_uint64()
        self.ctx.lib.rt_multi_frame_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        self.ctx.check(self.ctx.lib.rt_multi_frame_times(self.h, C.byref(a), C.byref(g), C.byref(n)))
        return dict(trace_ms=a.value, gather_ms=g.value, frames=n.value)

    def layout(self) -> dict:
        v = [C.c_uint32() for _ in range(4)]
        self.ctx.check(self.ctx.lib.rt_multi_frame_layout(self.h, *[C.byref(x) for x in v]))
        return dict(tiling=v[0].value, own_slots=v[1].value, root_slots=v[2].value, peer_slots=v[3].value)

    def wait(self):
        self.ctx.check(self.ctx.lib.rt_multi_frame_wait(self.h))

    def pixels_ptr(self) -> int:
        return int(self.ctx.lib.rt_multi_frame_pixels(self.h) or 0)

    def download_frame(self) -> np.ndarray:
        n = int(self.cam['resX'][0]) * int(self.cam['resY'][0])
        return self.ctx.download(self.pixels_ptr(), n * 4, np.float32).reshape(n, 4)

    def close(self):
        if self.h:
            self.ctx.lib.rt_multi_frame_destroy(self.h)
            self.h = None


def multi_frame_render(frames) -> None:
    """one frame of the given ranks: [frame] with one process per GPU, all ranks in rank order for a single-process group"""
    n = len(frames)
    arr = (C.c_void_p * n)(*[f.h for f in frames])
    ctx = frames[0].ctx
    if ctx.lib.rt_multi_frame_render(arr, n):
        for f in frames:
            msg = f.ctx.lib.rt_last_error(f.ctx.h)
            if msg:
                raise RtError(msg.decode())
        raise RtError('rt_multi_frame_render failed')
