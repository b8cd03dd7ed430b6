"""ref_cl.py -- TEST INFRASTRUCTURE ONLY: runs the REFERENCE's own OpenCL C kernels on the MI355X.

The reference's device code (`src/CLRayTracerEngine/*.cl` + the device branch of `Include/CLData/**`) needs
neither Boost nor Windows: ROCm clang's OpenCL C front end compiles it for gfx950 as it lies under
/root/reference (`make -C oracle ref_cl`; the only non-default switch the sources need is -fms-extensions, which
makes clang accept their back-slash include paths).  The resulting code objects live in oracle/_ref/cl_*/ (git-ignored,
shipped to the GPU box like any built .so) and are loaded here through the HIP module API -- no OpenCL runtime, no
stand-in headers, nothing of the reference is copied into the repository.

What this file restates is only the HOST side that cannot be compiled (Boost / Windows.h): buffer sizing and launch
shapes of BVHManager.cpp:122-329, BitonicSort.cpp:118-189, PrefixSum.cpp:94-210 and TwoLevelGridManager.cpp:207-513.

Two builds (oracle/Makefile):
  strict   -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt : IEEE fp32 divide / sqrt and no fused multiply-add,
           i.e. the arithmetic of the headers' host branch, which the oracle restates.  normalize() is still the device
           branch's fast_normalize (CLPortability.h:412).
  default  the options the reference passes to clBuildProgram ("-I <headers>", BVHManager.cpp:86): the compiler may fuse
           and use approximate division, as OpenCL allows.

SAFETY: the reference kernels have no bounds checks (private stack[32], BVH.h:349; counters + (parent - leafCount),
BVHKernels.cl:70).  Callers must hand over inputs on which every index stays in range -- see the guards in
tests/test_gpu_reference_cl.py -- a GPU memory fault takes the whole box down.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, '_ref')
WARP = 64          # CL_KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE on AMD (CLExecutionContext.cpp:522)
MAX_WG = 256       # CL_DEVICE_MAX_WORK_GROUP_SIZE on AMD

NODE = np.dtype([('data', '<u4', 4), ('lo', '<f4', 4), ('hi', '<f4', 4)])
PAIR = np.dtype([('key', '<u4'), ('value', '<u4')])
CONTACT = np.dtype([('pixelIndex', '<u4'), ('materialIndex', '<u4'), ('pad', '<u4', 2), ('normal_t', '<f4', 4)])
RAY = np.dtype([('idx', '<u4'), ('pad', '<u4', 3), ('origin', '<f4', 4), ('direction', '<f4', 4)])

MODULES = ('BVHKernels', 'SortKernels', 'PrefixSumKernels', 'TwoLevelGridKernels')


def available(variant: str = 'strict') -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, 'cl_' + variant, m + '.co')) for m in MODULES)


def closest_multiple_to(number: int, multiple: int) -> int:      # RTKernelUtils.h:175-181
    return number if number % multiple == 0 else (number // multiple + 1) * multiple


class HipError(RuntimeError):
    pass


class _Hip:
    def __init__(self):
        try:
            self.l = C.CDLL('libamdhip64.so.7')
        except OSError:
            self.l = C.CDLL('/opt/rocm/lib/libamdhip64.so')
        L = self.l
        L.hipGetErrorString.restype = C.c_char_p
        L.hipGetErrorString.argtypes = [C.c_int]
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        L.hipModuleUnload.argtypes = [C.c_void_p]
        L.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
        L.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.POINTER(C.c_void_p),
                                                                           C.POINTER(C.c_void_p)]

    def check(self, rc, what):
        if rc != 0:
            raise HipError('%s: %s' % (what, self.l.hipGetErrorString(rc).decode()))


class RefCL:
    """The reference's kernels, launched with the reference's launch shapes."""

    def __init__(self, variant: str = 'strict', device: int = 0):
        if not available(variant):
            raise FileNotFoundError('oracle/_ref/cl_%s/*.co missing: run `make -C oracle ref_cl` where /root/reference exists' % variant)
        self.hip = _Hip()
        self.hip.check(self.hip.l.hipSetDevice(device), 'hipSetDevice')
        self.mods = {}
        self.funcs = {}
        for m in MODULES:
            h = C.c_void_p()
            self.hip.check(self.hip.l.hipModuleLoad(C.byref(h), os.path.join(REF_DIR, 'cl_' + variant, m + '.co').encode()),
                           'hipModuleLoad ' + m)
            self.mods[m] = h
        self._live = []

    def close(self):
        for p in self._live:
            self.hip.l.hipFree(p)
        self._live = []
        for h in self.mods.values():
            self.hip.l.hipModuleUnload(h)
        self.mods = {}

    # ---- device memory ---------------------------------------------------------------------------------------
    def alloc(self, nbytes: int, fill: int | None = 0) -> C.c_void_p:
        p = C.c_void_p()
        self.hip.check(self.hip.l.hipMalloc(C.byref(p), max(int(nbytes), 16)), 'hipMalloc')
        if fill is not None:
            self.hip.check(self.hip.l.hipMemset(p, fill, max(int(nbytes), 16)), 'hipMemset')
        self._live.append(p)
        return p

    def upload(self, a) -> C.c_void_p:
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes, None)
        if a.nbytes:
            self.hip.check(self.hip.l.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), 'hipMemcpy H2D')
        return p

    def download(self, p, dtype, count) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        if out.nbytes:
            self.hip.check(self.hip.l.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2), 'hipMemcpy D2H')
        return out

    def free(self, p):
        self.hip.l.hipFree(p)
        self._live = [q for q in self._live if q.value != p.value]

    # ---- launch ----------------------------------------------------------------------------------------------
    def launch(self, module: str, name: str, global_size: int, local_size: int, args, lds_bytes: int = 0):
        """clEnqueueNDRangeKernel(global, local) + clFlush + clWaitForEvents (e.g. BVHManager.cpp:184-192).
        args: ctypes values in kernel-argument order (c_void_p for buffers; a __local pointer argument is passed as
        the c_uint32 LDS byte offset of the buffer, with lds_bytes the dynamic LDS the launch needs)."""
        assert global_size % local_size == 0 and global_size > 0
        key = (module, name)
        if key not in self.funcs:
            f = C.c_void_p()
            self.hip.check(self.hip.l.hipModuleGetFunction(C.byref(f), self.mods[module], name.encode()), 'hipModuleGetFunction ' + name)
            self.funcs[key] = f
        holders = list(args)
        params = (C.c_void_p * len(holders))(*[C.cast(C.byref(h), C.c_void_p) for h in holders])
        self.hip.check(self.hip.l.hipModuleLaunchKernel(self.funcs[key], global_size // local_size, 1, 1, local_size, 1, 1,
                                                         lds_bytes, None, params, None), 'launch ' + name)
        self.hip.check(self.hip.l.hipDeviceSynchronize(), 'run ' + name)

    # ---- BVHManager::initializeFrame + construct, stage by stage (BVHManager.cpp:122-228) ----------------------
    @staticmethod
    def morton_buffer_items(n: int) -> int:                   # BVHManager.cpp:128-131
        p = 1 << (int(n).bit_length() - 1)
        return p * 2 if p < n else p

    def calculate_morton_codes(self, scene: bytes, n: int):
        """kernel calculateMortonCodes (BVHKernels.cl:46-50) on zero-filled nodes and UINT_MAX-filled pairs
        (BVHManager.cpp:134-142).  Returns (nodes[2n-1], pairs[N])."""
        N = self.morton_buffer_items(n)
        d_nodes = self.alloc((2 * n - 1) * 48, 0)
        d_pairs = self.alloc(N * 8, 0xFF)
        d_scene = self.upload(np.frombuffer(scene, dtype=np.uint8))
        self.launch('BVHKernels', 'calculateMortonCodes', closest_multiple_to(n, WARP), WARP, [d_nodes, d_pairs, d_scene])
        nodes, pairs = self.download(d_nodes, NODE, 2 * n - 1), self.download(d_pairs, PAIR, N)
        for p in (d_nodes, d_pairs, d_scene):
            self.free(p)
        return nodes, pairs

    def bitonic_sort(self, pairs: np.ndarray) -> np.ndarray:
        """BitonicSort::sort (BitonicSort.cpp:118-189), ALLOWB = 2+4+8: kernels ParallelBitonic_B8 / _B4 / _B2."""
        num_items = len(pairs)
        assert num_items and num_items & (num_items - 1) == 0
        d = self.upload(pairs)
        launches = 0
        length = 1
        while length < num_items:
            inc = length
            while inc > 0:
                if inc >= 4:
                    kernel, ninc = 'ParallelBitonic_B8', 3
                elif inc >= 2:
                    kernel, ninc = 'ParallelBitonic_B4', 2
                else:
                    kernel, ninc = 'ParallelBitonic_B2', 1
                n_threads = num_items >> ninc
                wg = min(MAX_WG, 256, n_threads)
                if n_threads == 0:                  # num_items < 2^ninc cannot happen: inc >= 2^(ninc-1) and inc < num_items
                    raise AssertionError
                self.launch('SortKernels', kernel, n_threads, wg, [d, C.c_int(inc), C.c_int(length << 1)])
                launches += 1
                inc >>= ninc
            length <<= 1
        out = self.download(d, PAIR, num_items)
        self.free(d)
        self.sort_launches = launches
        return out

    def build_radix_tree(self, nodes: np.ndarray, sorted_pairs: np.ndarray, n: int) -> np.ndarray:
        """kernel buildRadixTree (BVHKernels.cl:55-60), LITERAL: its guard idx < leafCount-2 leaves inner node n-2 alone."""
        d_nodes, d_pairs = self.upload(nodes), self.upload(sorted_pairs)
        self.launch('BVHKernels', 'buildRadixTree', closest_multiple_to(n - 1, WARP), WARP, [d_nodes, d_pairs, C.c_uint32(n)])
        out = self.download(d_nodes, NODE, 2 * n - 1)
        self.free(d_nodes); self.free(d_pairs)
        return out

    def compute_bounding_boxes(self, nodes: np.ndarray, n: int) -> np.ndarray:
        """kernel computeBoundingBoxes (BVHKernels.cl:65-79).  Every parent link of `nodes` must be either UINT_MAX or in
        [n, 2n-1): the kernel does atomic_inc(counters + (parent - leafCount)) unchecked."""
        par = nodes['data'][:, 0]
        ok = (par == 0xFFFFFFFF) | ((par >= n) & (par < 2 * n - 1))
        if not bool(ok.all()):
            raise ValueError('computeBoundingBoxes would index its counters out of range')
        d_nodes, d_counters = self.upload(nodes), self.alloc(n * 4, 0)
        self.launch('BVHKernels', 'computeBoundingBoxes', closest_multiple_to(n, WARP), WARP, [d_nodes, d_counters, C.c_uint32(n)])
        out = self.download(d_nodes, NODE, 2 * n - 1)
        self.free(d_nodes); self.free(d_counters)
        return out

    # ---- BVHManager::generateContacts, both overloads (BVHManager.cpp:234-329) ---------------------------------
    def generate_contacts(self, cam: np.ndarray, nodes: np.ndarray, n: int, scene: bytes) -> np.ndarray:
        count = int(cam['resX'][0]) * int(cam['resY'][0])
        d_cam, d_nodes = self.upload(cam), self.upload(nodes)
        d_scene = self.upload(np.frombuffer(scene, dtype=np.uint8))
        d_out = self.alloc(count * 32, 0)
        self.launch('BVHKernels', 'generateContacts', closest_multiple_to(count, WARP), WARP,
                    [d_cam, d_nodes, C.c_uint32(n), d_scene, d_out])
        out = self.download(d_out, CONTACT, count)
        for p in (d_cam, d_nodes, d_scene, d_out):
            self.free(p)
        return out

    def generate_contacts2(self, rays: np.ndarray, nodes: np.ndarray, n: int, scene: bytes) -> np.ndarray:
        count = len(rays)
        d_rays, d_nodes = self.upload(rays), self.upload(nodes)
        d_scene = self.upload(np.frombuffer(scene, dtype=np.uint8))
        d_out = self.alloc(count * 32, 0)
        self.launch('BVHKernels', 'generateContacts2', closest_multiple_to(count, WARP), WARP,
                    [d_rays, C.c_uint32(count), d_nodes, C.c_uint32(n), d_scene, d_out])
        out = self.download(d_out, CONTACT, count)
        for p in (d_rays, d_nodes, d_scene, d_out):
            self.free(p)
        return out

    # ---- PrefixSum::computePrefixSum (PrefixSum.cpp:94-133) ----------------------------------------------------
    def prefix_sum_one_group(self, values: np.ndarray) -> np.ndarray:
        """kernel group_prefixSum (PrefixSumKernels.cl:28-86) as PrefixSum::invokeGroupKernel launches it for offset 1,
        restricted to size <= 2 * maxWorkgroupSize: for larger inputs the kernel's tree loops run over the TOTAL length
        (`for (l = size >> 1; ...)`, :46) and index its __local block out of range, so only the in-range case is run here."""
        size = len(values)
        assert size >= 2 and size & (size - 1) == 0 and size <= 2 * MAX_WG
        d_in, d_out = self.upload(values.astype(np.uint32)), self.alloc(size * 4, 0)
        local_threads = MAX_WG
        global_threads = ((((size + 1) >> 1) + local_threads - 1) // local_threads) * local_threads
        lds = (local_threads * 4) << 1
        # the __local pointer argument: byte offset 0 of the launch's dynamic LDS (the kernel has no static LDS)
        self.launch('PrefixSumKernels', 'group_prefixSum', global_threads, local_threads,
                    [d_out, d_in, C.c_int(1), C.c_int(size), C.c_uint32(0)], lds_bytes=lds)
        out = self.download(d_out, np.uint32, size)
        self.free(d_in); self.free(d_out)
        return out

    # ---- TwoLevelGridManager (TwoLevelGridManager.cpp:207-551), the kernels that stay in range on a consistent grid ----
    def tlg_prepare_data(self, scene: bytes, grid: np.ndarray, n: int) -> np.ndarray:
        """kernel prepareDataKernel (TwoLevelGridKernels.cl:43-51): top-level cells overlapped per triangle; the kernel
        skips the last triangle (`tris = total - 1`), whose counter keeps the 0 of initializeFrame."""
        d_scene = self.upload(np.frombuffer(scene, dtype=np.uint8))
        d_grid, d_counters = self.upload(grid), self.alloc(n * 4, 0)
        self.launch('TwoLevelGridKernels', 'prepareDataKernel', closest_multiple_to(n, WARP), WARP, [d_scene, d_grid, d_counters])
        out = self.download(d_counters, np.uint32, n)
        for p in (d_scene, d_grid, d_counters):
            self.free(p)
        return out

    def tlg_fill_top_level_cells(self, ranges: np.ndarray, grid: np.ndarray):
        """kernel countLeavesAndFillCellKernel (TwoLevelGridKernels.cl:121-126) -> fillTopLevelCell / calcLeafCellResolution
        (TwoLevelGrid.h:190-227): leaf resolution floor(step * pow(density * k / V, oneThird)) with the DEVICE's pow."""
        cells = len(ranges)
        d_ranges, d_grid = self.upload(ranges), self.upload(grid)
        d_counters, d_cells = self.alloc(cells * 4, 0), self.alloc(cells * 16, 0)
        self.launch('TwoLevelGridKernels', 'countLeavesAndFillCellKernel', closest_multiple_to(cells, WARP), WARP,
                    [d_ranges, d_counters, d_cells, C.c_uint32(cells), d_grid])
        top = self.download(d_cells, np.dtype([('res', '<u4', 3), ('firstLeafIdx', '<u4')]), cells)
        counts = self.download(d_counters, np.uint32, cells)
        for p in (d_ranges, d_grid, d_counters, d_cells):
            self.free(p)
        return top, counts

    def tlg_generate_contacts2(self, rays: np.ndarray, scene: bytes, grid: np.ndarray, top_cells: np.ndarray,
                               leaf_ranges: np.ndarray, leaf_pairs: np.ndarray) -> np.ndarray:
        """kernel generateContacts2Kernel (TwoLevelGridKernels.cl:251-270) -> tlg_generate_contact (TwoLevelGrid.h:557-678).
        Rays must have finite, non-zero direction components: the nested DDA's `while (minimal != next[axis]) axis++`
        (:513, :640) never ends on NaN."""
        d = rays['direction'][:, :3]
        if not bool((np.isfinite(d) & (d != 0)).all()):
            raise ValueError('the reference DDA does not terminate on zero / non-finite direction components')
        count = len(rays)
        d_rays = self.upload(rays)
        d_scene = self.upload(np.frombuffer(scene, dtype=np.uint8))
        d_grid, d_top = self.upload(grid), self.upload(top_cells)
        d_lr, d_lp = self.upload(leaf_ranges), self.upload(leaf_pairs)
        d_out = self.alloc(count * 32, 0)
        self.launch('TwoLevelGridKernels', 'generateContacts2Kernel', closest_multiple_to(count, WARP), WARP,
                    [d_rays, C.c_uint32(count), d_scene, d_grid, d_top, d_lr, d_lp, d_out])
        out = self.download(d_out, CONTACT, count)
        for p in (d_rays, d_scene, d_grid, d_top, d_lr, d_lp, d_out):
            self.free(p)
        return out
