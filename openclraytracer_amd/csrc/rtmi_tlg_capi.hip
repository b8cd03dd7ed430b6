// rtmi_tlg_capi.hip -- C ABI of the Two-Level Grid manager (include/rtmi.h, rt_tlg_*): the host logic of
// AccelerationStructures::TwoLevelGridManager (src/CLRayTracerEngine/TwoLevelGridManager.cpp:60-600) on the HIP kernels
// of tlg_kernels.hip and the scan / bitonic-network primitives of sort_scan_kernels.hip.
#include "rtmi_internal.h"
#include "tlg_math.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

// largestPowerOfTwo(x) << 1 (RTKernelUtils.h:160-164, TwoLevelGridManager.cpp:165,167,239,398): always > x
static uint32_t pow2_above(uint32_t x)
{
    if (x == 0) return 1;
    uint64_t p = 1;
    while (p * 2 <= x) p *= 2;
    return (uint32_t)(p * 2);
}

int rt_tlg_create(rt_ctx* ctx, const rt_scene* scene, rt_tlg** out)
{
    if (!ctx || !scene || !out) return RT_FAIL(ctx, "null argument");
    *out = nullptr;
    if (scene->ctx != ctx) return RT_FAIL(ctx, "scene belongs to another context");
    rt_tlg* g = new (std::nothrow) rt_tlg();
    if (!g) return RT_FAIL(ctx, "out of host memory");
    g->ctx = ctx; g->scene = scene;
    *out = g;
    return RT_SUCCESS;
}

#define TLG_AXIS_MAX_HOST 512u            // = TLG_AXIS_MAX of tlg_kernels.hip: planes the trace kernel's LDS table holds
static_assert(sizeof(TlgCamPlane) == sizeof(TlgTopRec), "the camera planes live behind the cell records");
static void free_tlg_buffers(rt_tlg* g)
{
    void** ptrs[] = { (void**)&g->d_counters, (void**)&g->d_scan, (void**)&g->d_scan_scratch, (void**)&g->d_cell_ranges, (void**)&g->d_top,
                      (void**)&g->d_pairs, (void**)&g->d_leaf_pairs, (void**)&g->d_leaf_ranges, (void**)&g->d_written, (void**)&g->d_tricam,
                      (void**)&g->d_toprec, (void**)&g->d_axis_tab, (void**)&g->d_leafrec };
    for (void** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    g->counters_cap = 0; g->pairs_cap = 0; g->leaf_cap = 0; g->leaf_ranges_cap = 0; g->tricam_valid = false; g->axis_cap = 0; g->leafrec_cap = 0; g->walk_records = false;
}

void rt_tlg_destroy(rt_tlg* g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    free_tlg_buffers(g);
    delete g;
}

int rt_tlg_set_densities(rt_tlg* g, float top_level, float leaf)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    if (!(top_level > 0.0f) || !(leaf > 0.0f)) return RT_FAIL(g->ctx, "densities must be positive");
    g->top_density = top_level; g->leaf_density = leaf;
    return RT_SUCCESS;
}

// calculateGridData + getResolution, TwoLevelGridManager.cpp:566-600
static int calculate_grid_data(rt_tlg* g)
{
    const DeviceScene& sc = g->scene->d;
    rt_grid_data& G = g->grid;
    std::memset(&G, 0, sizeof G);
    for (int k = 0; k < 3; k++) { G.box_lo[k] = sc.bb_lo[k]; G.box_hi[k] = sc.bb_hi[k]; }
    G.box_lo[3] = g->scene->bb_w[0]; G.box_hi[3] = g->scene->bb_w[1];
    const float deltaX = G.box_hi[0] - G.box_lo[0], deltaY = G.box_hi[1] - G.box_lo[1], deltaZ = G.box_hi[2] - G.box_lo[2];
    const float volume = deltaX * deltaY * deltaZ;                        // boxVolume, AABB.h:243-249
    const float prims = (float)sc.n_tris;
    const float a = tlg_pow_third(g->top_density * prims / volume);
    G.resX = tlg_f2u(deltaX * a); G.resY = tlg_f2u(deltaY * a); G.resZ = tlg_f2u(deltaZ * a);
    if (G.resX == 0 || G.resY == 0 || G.resZ == 0)
        return RT_FAIL(g->ctx, "scene bounding box is degenerate: the grid resolution has a zero axis");
    if ((uint64_t)G.resX * G.resY * G.resZ > 0x40000000ull) return RT_FAIL(g->ctx, "grid resolution too large");
    G.stepX = (G.box_hi[0] - G.box_lo[0]) / (float)G.resX;
    G.stepY = (G.box_hi[1] - G.box_lo[1]) / (float)G.resY;
    G.stepZ = (G.box_hi[2] - G.box_lo[2]) / (float)G.resZ;
    G.leafDensity = g->leaf_density;
    return RT_SUCCESS;
}

static int ensure_counters(rt_tlg* g, uint64_t words)
{
    rt_ctx* ctx = g->ctx;
    if (words <= g->counters_cap) return RT_SUCCESS;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (g->d_counters) { (void)hipFree(g->d_counters); g->d_counters = nullptr; }
    if (g->d_scan) { (void)hipFree(g->d_scan); g->d_scan = nullptr; }
    if (g->d_scan_scratch) { (void)hipFree(g->d_scan_scratch); g->d_scan_scratch = nullptr; }
    g->counters_cap = 0;
    RT_HIP(ctx, hipMalloc((void**)&g->d_counters, words * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void**)&g->d_scan, words * sizeof(uint32_t)));
    RT_HIP(ctx, hipMalloc((void**)&g->d_scan_scratch, scan_scratch_words((uint32_t)words) * sizeof(uint32_t)));
    g->counters_cap = words;
    return RT_SUCCESS;
}

// what initializeFrame leaves behind (TwoLevelGridManager.cpp:181-199): counters, scan output, cell ranges and top-level cells zeroed
static int clear_frame_buffers(rt_tlg* g)
{
    rt_ctx* ctx = g->ctx;
    RT_HIP(ctx, hipMemsetAsync(g->d_counters, 0, g->counters_cap * sizeof(uint32_t), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(g->d_scan, 0, g->counters_cap * sizeof(uint32_t), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(g->d_cell_ranges, 0, (size_t)g->info.cells * sizeof(uint2), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(g->d_top, 0, (size_t)g->info.cells * sizeof(rt_top_cell), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(g->d_written, 0, 256, ctx->stream));
    return RT_SUCCESS;
}

int rt_tlg_initialize_frame(rt_tlg* g)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    RT_ENTER(ctx);
    g->n = g->scene->d.n_tris;
    if (g->n == 0) return RT_FAIL(ctx, "the scene has no triangles");
    if (calculate_grid_data(g)) return RT_ERROR;
    const uint32_t cells = g->grid.resX * g->grid.resY * g->grid.resZ;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cells != g->info.cells || !g->d_top) {
        if (g->d_cell_ranges) { (void)hipFree(g->d_cell_ranges); g->d_cell_ranges = nullptr; }
        if (g->d_top) { (void)hipFree(g->d_top); g->d_top = nullptr; }
        g->cells_cap = 0;
        RT_HIP(ctx, hipMalloc((void**)&g->d_cell_ranges, (size_t)cells * sizeof(uint2)));
        RT_HIP(ctx, hipMalloc((void**)&g->d_top, (size_t)cells * sizeof(rt_top_cell)));
        g->cells_cap = cells;
        // the walk's own records of the cells (tlg_kernels.hip, k_tlg_trace_records): without them (allocation failed, knob RTMI_TLG_WALK_RECORDS=0,
        // more planes than the kernel's table holds) the grid is walked as in round 4
        if (g->d_toprec) { (void)hipFree(g->d_toprec); g->d_toprec = nullptr; }
        if (!ctx->tlg_no_walk_records && hipMalloc((void**)&g->d_toprec, ((size_t)cells + TLG_AXIS_MAX_HOST) * sizeof(TlgTopRec)) != hipSuccess) { (void)hipGetLastError(); g->d_toprec = nullptr; }
    }
    {
        const uint32_t planes = g->grid.resX + g->grid.resY + g->grid.resZ;
        if (g->d_toprec && planes > g->axis_cap) {
            if (g->d_axis_tab) { (void)hipFree(g->d_axis_tab); g->d_axis_tab = nullptr; g->axis_cap = 0; }
            if (hipMalloc((void**)&g->d_axis_tab, (size_t)planes * sizeof(float4)) != hipSuccess) { (void)hipGetLastError(); g->d_axis_tab = nullptr; }
            else g->axis_cap = planes;
        }
    }
    g->walk_records = false;
    if (!g->d_written) RT_HIP(ctx, hipMalloc((void**)&g->d_written, 256));
    g->info = rt_tlg_info{};
    g->info.cells = cells;
    if (ensure_counters(g, g->n > cells ? g->n : cells)) return RT_ERROR;
    if (clear_frame_buffers(g)) return RT_ERROR;
    g->frame_initialised = true;
    g->built = false;
    return RT_SUCCESS;
}

// one 4-byte read-back of the last element of a scan (TwoLevelGridManager.cpp:237,345,393)
static int read_word(rt_ctx* ctx, const uint32_t* d_src, uint32_t* out)
{
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(out, d_src, 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_SUCCESS;
}

// The last launch of a build: the walk's records from the finished top-level cells (flag word d_written[8]: a plane or a leaf grid outside
// the bounds of the short division sequences).  false: no records for this build.
static bool enqueue_walk_records(rt_tlg* g)
{
    const uint32_t planes = g->grid.resX + g->grid.resY + g->grid.resZ;
    if (!g->d_toprec || !g->d_axis_tab || planes > g->axis_cap || planes > TLG_AXIS_MAX_HOST) return false;
    // (the walk with the records addresses its buffers by 32-bit byte offsets: each of them below 4 GB)
    if (((uint64_t)g->info.cells + TLG_AXIS_MAX_HOST) * sizeof(TlgTopRec) >= (1ull << 32) || (uint64_t)g->leaf_ranges_cap * sizeof(uint2) >= (1ull << 32) ||
        (uint64_t)g->leaf_cap * sizeof(uint2) >= (1ull << 32) || (uint64_t)g->n * sizeof(TriCam) >= (1ull << 32)) return false;
    const float* lo = g->grid.box_lo; const float* hi = g->grid.box_hi;
    for (int k = 0; k < 3; k++) if (!(std::fabs(lo[k]) <= 0x1p30f) || !(std::fabs(hi[k]) <= 0x1p30f)) return false;
    // the leaf-cell records cover the leaf-range array as allocated (ranges beyond this build's leaf cells are zero: empty records)
    if (g->leafrec_cap != g->leaf_ranges_cap || !g->d_leafrec) {
        (void)hipStreamSynchronize(g->ctx->stream);
        if (g->d_leafrec) { (void)hipFree(g->d_leafrec); g->d_leafrec = nullptr; g->leafrec_cap = 0; }
        if ((uint64_t)g->leaf_ranges_cap * 2u * sizeof(uint4) >= (1ull << 32)) return false;
        if (hipMalloc((void**)&g->d_leafrec, (size_t)g->leaf_ranges_cap * 2u * sizeof(uint4)) != hipSuccess) { (void)hipGetLastError(); g->d_leafrec = nullptr; return false; }
        g->leafrec_cap = g->leaf_ranges_cap;
    }
    launch_tlg_leaf_records(g->ctx->stream, g->d_leaf_ranges, g->d_leaf_pairs, g->leaf_ranges_cap, g->leaf_cap, g->d_leafrec);
    launch_tlg_trace_records(g->ctx->stream, g->grid, g->d_top, g->info.cells, g->d_toprec, g->d_axis_tab, g->d_written + 8);
    return true;
}

// The reference's sequence with its three 4-byte read-backs: they size the pair arrays, and the sizes of the bitonic
// networks (powers of two above the counts) are part of the result -- the network's extra levels permute equal keys.
// Used by the first build and whenever the sizes of the previous build no longer fit.
static int construct_with_readbacks(rt_tlg* g)
{
    rt_ctx* ctx = g->ctx;
    hipStream_t s = ctx->stream;
    const DeviceScene& sc = g->scene->d;
    const rt_grid_data& G = g->grid;
    const uint32_t n = g->n, cells = g->info.cells;

    // 1. cells overlapped by every triangle's box, scan, pair count                               (:213-238)
    launch_tlg_count_top(s, G, sc, g->d_counters);
    launch_scan_inclusive(s, g->d_counters, g->d_scan, n, g->d_scan_scratch);
    uint32_t pairs = 0;
    if (read_word(ctx, g->d_scan + (n - 1), &pairs)) return RT_ERROR;
    if (pairs == 0) return RT_FAIL(ctx, "no (cell, triangle) pairs: nothing to build");
    if (pairs > 0x3FFFFFFFu) return RT_FAIL(ctx, "too many (cell, triangle) pairs");
    const uint32_t pairs_pow2 = pow2_above(pairs);
    if (pairs_pow2 != g->pairs_cap || !g->d_pairs) {
        if (g->d_pairs) { (void)hipFree(g->d_pairs); g->d_pairs = nullptr; g->pairs_cap = 0; }
        RT_HIP(ctx, hipMalloc((void**)&g->d_pairs, (size_t)pairs_pow2 * sizeof(uint2)));
        g->pairs_cap = pairs_pow2;
    }
    g->info.pairs_count = pairs; g->info.pairs_pow2 = pairs_pow2;
    RT_HIP(ctx, hipMemsetAsync(g->d_pairs, 0xFF, (size_t)pairs_pow2 * sizeof(uint2), s));         // :248-250
    // 2. emit (cell, triangle), sort by cell                                                        (:252-267)
    launch_tlg_emit_top(s, G, sc, g->d_scan, g->d_counters, g->d_pairs, pairs_pow2);
    launch_bitonic_sort(s, g->d_pairs, pairs_pow2);
    // 3. [first, end) of every top-level cell, leaf resolution, leaf-cell count                    (:269-346)
    launch_segment_bounds(s, g->d_pairs, pairs, nullptr, g->d_cell_ranges, cells);
    RT_HIP(ctx, hipMemsetAsync(g->d_counters, 0, g->counters_cap * sizeof(uint32_t), s));
    RT_HIP(ctx, hipMemsetAsync(g->d_scan, 0, g->counters_cap * sizeof(uint32_t), s));
    launch_tlg_cell_res(s, G, g->d_cell_ranges, cells, g->d_top, g->d_counters);
    launch_scan_inclusive(s, g->d_counters, g->d_scan, cells, g->d_scan_scratch);
    uint32_t leaf_cells = 0;
    if (read_word(ctx, g->d_scan + (cells - 1), &leaf_cells)) return RT_ERROR;
    launch_tlg_first_leaf(s, g->d_top, g->d_scan, cells);                                           // :347-360
    g->info.leaf_cells_count = leaf_cells;
    // 4. leaf cells overlapped per top-level pair, scan, leaf pair count                            (:362-395)
    if (ensure_counters(g, pairs_pow2)) return RT_ERROR;          // (the capacity, so that a later rebuild can scan over it)
    launch_tlg_count_leaf(s, G, g->d_pairs, pairs, nullptr, g->d_top, sc, g->d_counters);
    launch_scan_inclusive(s, g->d_counters, g->d_scan, pairs, g->d_scan_scratch);
    uint32_t leaf_pairs = 0;
    if (read_word(ctx, g->d_scan + (pairs - 1), &leaf_pairs)) return RT_ERROR;
    if (leaf_pairs > 0x3FFFFFFFu) return RT_FAIL(ctx, "too many (leaf cell, triangle) pairs");
    const uint32_t leaf_pow2 = pow2_above(leaf_pairs);
    if (leaf_pow2 != g->leaf_cap || !g->d_leaf_pairs) {
        if (g->d_leaf_pairs) { (void)hipFree(g->d_leaf_pairs); g->d_leaf_pairs = nullptr; g->leaf_cap = 0; }
        RT_HIP(ctx, hipMalloc((void**)&g->d_leaf_pairs, (size_t)leaf_pow2 * sizeof(uint2)));
        g->leaf_cap = leaf_pow2;
    }
    const uint32_t want_ranges = leaf_cells ? leaf_cells : 1u;
    if (want_ranges > g->leaf_ranges_cap || !g->d_leaf_ranges) {
        if (g->d_leaf_ranges) { (void)hipFree(g->d_leaf_ranges); g->d_leaf_ranges = nullptr; g->leaf_ranges_cap = 0; }
        const uint32_t cap = want_ranges + want_ranges / 8u + 64u;   // head room: a rebuild of a slightly changed scene still fits
        RT_HIP(ctx, hipMalloc((void**)&g->d_leaf_ranges, (size_t)cap * sizeof(uint2)));
        g->leaf_ranges_cap = cap;
    }
    g->info.leaf_pairs_count = leaf_pairs; g->info.leaf_pairs_pow2 = leaf_pow2;
    RT_HIP(ctx, hipMemsetAsync(g->d_leaf_pairs, 0xFF, (size_t)leaf_pow2 * sizeof(uint2), s));     // :404-408
    RT_HIP(ctx, hipMemsetAsync(g->d_leaf_ranges, 0, (size_t)g->leaf_ranges_cap * sizeof(uint2), s));
    // 5. emit (leaf cell, triangle) with the exact triangle / box test, sort, leaf ranges          (:410-466)
    launch_tlg_emit_leaf(s, G, g->d_pairs, pairs, nullptr, g->d_top, sc, g->d_scan, g->d_counters, g->d_leaf_pairs, leaf_pow2, g->d_written);
    launch_bitonic_sort(s, g->d_leaf_pairs, leaf_pow2);
    launch_segment_bounds(s, g->d_leaf_pairs, leaf_pairs, nullptr, g->d_leaf_ranges, leaf_cells);
    const bool records = enqueue_walk_records(g);
    uint32_t h[12] = { 0 };
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(h, g->d_written, sizeof h, hipMemcpyDeviceToHost, s));
    RT_HIP(ctx, hipStreamSynchronize(s));                                // construct() returns with the build finished
    g->info.written_leaf_pairs = h[0];
    g->walk_records = records && h[8] == 0u;
    return RT_SUCCESS;
}

// A REBUILD (dynamic scenes rebuild per frame, OpenCLRayTracerDemo.cpp:230-234) runs without host round trips: every kernel
// is enqueued on the array sizes of the previous build, the counts stay on the device (three words behind d_written; the
// kernels that need a count read it there), and ONE read-back at the end -- which construct() needs anyway, it returns with
// the build finished -- tells whether the sizes still fit: the same powers of two above the counts (they are part of the
// result, see above) and a leaf-range array that is large enough.  If so the products are exactly those of the sequence
// above; if not, that sequence runs.  Writes never pass a capacity either way.
static int construct_on_previous_sizes(rt_tlg* g, bool* accepted)
{
    rt_ctx* ctx = g->ctx;
    hipStream_t s = ctx->stream;
    const DeviceScene& sc = g->scene->d;
    const rt_grid_data& G = g->grid;
    const uint32_t n = g->n, cells = g->info.cells, P = g->pairs_cap, L = g->leaf_cap;
    uint32_t* d_cnt = g->d_written + 4;                                   // pairs, leaf cells, leaf pairs
    *accepted = false;
    // Every kernel below is enqueued on a CAPACITY, not on a count the host has seen: each of them must belong to the arrays as they
    // are now.  (Round 2, 09:07: the counters / scan arrays were sized for the pair COUNT of the build with read-backs while this
    // sequence zeroes counters[count .. P) (k_tlg_count_leaf) and scans over P, the power of two above it -- writes past the end of
    // d_counters / d_scan that went unnoticed until initializeFrame with other densities moved the allocations, then a GPU memory
    // fault inside construct().  Fixed by sizing them for P, `ensure_counters(g, pairs_pow2)`; asserted here for every array.)
    if (!(g->counters_cap >= n && g->counters_cap >= cells && g->counters_cap >= P))
        return RT_FAIL(ctx, "internal: the counter / scan arrays are smaller than what a rebuild on the previous sizes scans over");
    if (!(cells && cells <= g->cells_cap && g->d_cell_ranges && g->d_top))
        return RT_FAIL(ctx, "internal: the cell arrays do not belong to this initializeFrame");
    if (!(P && L && g->d_pairs && g->d_leaf_pairs && g->d_leaf_ranges && g->leaf_ranges_cap && g->d_written))
        return RT_FAIL(ctx, "internal: a rebuild on the previous sizes without previous arrays");
    launch_tlg_count_top(s, G, sc, g->d_counters);
    launch_scan_inclusive(s, g->d_counters, g->d_scan, n, g->d_scan_scratch);
    RT_HIP(ctx, hipMemcpyAsync(d_cnt + 0, g->d_scan + (n - 1), 4, hipMemcpyDeviceToDevice, s));
    RT_HIP(ctx, hipMemsetAsync(g->d_pairs, 0xFF, (size_t)P * sizeof(uint2), s));
    launch_tlg_emit_top(s, G, sc, g->d_scan, g->d_counters, g->d_pairs, P);
    launch_bitonic_sort(s, g->d_pairs, P);
    launch_segment_bounds(s, g->d_pairs, P, d_cnt + 0, g->d_cell_ranges, cells);
    RT_HIP(ctx, hipMemsetAsync(g->d_counters, 0, g->counters_cap * sizeof(uint32_t), s));
    RT_HIP(ctx, hipMemsetAsync(g->d_scan, 0, g->counters_cap * sizeof(uint32_t), s));
    launch_tlg_cell_res(s, G, g->d_cell_ranges, cells, g->d_top, g->d_counters);
    launch_scan_inclusive(s, g->d_counters, g->d_scan, cells, g->d_scan_scratch);
    RT_HIP(ctx, hipMemcpyAsync(d_cnt + 1, g->d_scan + (cells - 1), 4, hipMemcpyDeviceToDevice, s));
    launch_tlg_first_leaf(s, g->d_top, g->d_scan, cells);
    launch_tlg_count_leaf(s, G, g->d_pairs, P, d_cnt + 0, g->d_top, sc, g->d_counters);       // zeros beyond the pairs
    launch_scan_inclusive(s, g->d_counters, g->d_scan, P, g->d_scan_scratch);                 // over the capacity: same prefix sums, total at the end
    RT_HIP(ctx, hipMemcpyAsync(d_cnt + 2, g->d_scan + (P - 1), 4, hipMemcpyDeviceToDevice, s));
    RT_HIP(ctx, hipMemsetAsync(g->d_leaf_pairs, 0xFF, (size_t)L * sizeof(uint2), s));
    RT_HIP(ctx, hipMemsetAsync(g->d_leaf_ranges, 0, (size_t)g->leaf_ranges_cap * sizeof(uint2), s));
    launch_tlg_emit_leaf(s, G, g->d_pairs, P, d_cnt + 0, g->d_top, sc, g->d_scan, g->d_counters, g->d_leaf_pairs, L, g->d_written);
    launch_bitonic_sort(s, g->d_leaf_pairs, L);
    launch_segment_bounds(s, g->d_leaf_pairs, L, d_cnt + 2, g->d_leaf_ranges, g->leaf_ranges_cap);
    const bool records = enqueue_walk_records(g);
    RT_HIP(ctx, hipGetLastError());
    uint32_t h[12] = { 0 };
    RT_HIP(ctx, hipMemcpyAsync(h, g->d_written, sizeof h, hipMemcpyDeviceToHost, s));
    RT_HIP(ctx, hipStreamSynchronize(s));
    const uint32_t pairs = h[4], leaf_cells = h[5], leaf_pairs = h[6];
    if (pairs == 0 || pairs > 0x3FFFFFFFu || leaf_pairs > 0x3FFFFFFFu) return RT_SUCCESS;      // let the full sequence report it
    if (pow2_above(pairs) != P || pow2_above(leaf_pairs) != L || leaf_cells > g->leaf_ranges_cap) return RT_SUCCESS;
    g->info.pairs_count = pairs; g->info.pairs_pow2 = P;
    g->info.leaf_cells_count = leaf_cells;
    g->info.leaf_pairs_count = leaf_pairs; g->info.leaf_pairs_pow2 = L;
    g->info.written_leaf_pairs = h[0];
    g->walk_records = records && h[8] == 0u;
    *accepted = true;
    return RT_SUCCESS;
}

int rt_tlg_construct(rt_tlg* g)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!g->frame_initialised) return RT_FAIL(ctx, "initializeFrame must precede construct");
    RT_ENTER(ctx);
    const bool allow = !ctx->tlg_no_rebuild_on_previous;          // knob RTMI_TLG_REBUILD_ON_PREVIOUS_SIZES=0
    bool done = false;
    if (allow && g->d_pairs && g->d_leaf_pairs && g->d_leaf_ranges && g->pairs_cap && g->leaf_cap && g->counters_cap >= g->pairs_cap) {
        if (construct_on_previous_sizes(g, &done)) return RT_ERROR;
        if (done) g->rebuilds_without_readback++;
        else if (clear_frame_buffers(g)) return RT_ERROR;         // the sizes changed: start over from what initializeFrame left
    }
    if (!done && construct_with_readbacks(g)) return RT_ERROR;
    g->frame_initialised = false;
    g->built = true;
    g->tricam_valid = false;                      // the scene's triangles may have moved
    return RT_SUCCESS;
}

static TlgLaunch make_tlg_launch(const rt_tlg* g)
{
    TlgLaunch t{};
    t.grid = g->grid; t.top = g->d_top; t.leaf_ranges = g->d_leaf_ranges; t.leaf_pairs = g->d_leaf_pairs; t.scene = &g->scene->d;
    if (g->walk_records) { t.toprec = g->d_toprec; t.axis_tab = g->d_axis_tab; t.leafrec = g->d_leafrec; }
    return t;
}

// Primary rays share their origin: what the triangle test computes from the origin and the triangle alone is computed once per
// camera position (TriCam, rtmi_internal.h), as for the BVH's packet kernel.  The table follows the camera and the scene
// (construct() invalidates it); without it (allocation failed, knob RTMI_TLG_TRICAM=0) the kernel reads the scene's table.
static void use_camera_triangles(rt_tlg* g, const rt_camera* cam, TlgLaunch& t)
{
    if (g->ctx->tlg_no_tricam || g->n == 0) return;
    if (g->d_tricam && g->tricam_n != g->n) {           // another scene size since: a table of that size
        (void)hipStreamSynchronize(g->ctx->stream);
        (void)hipFree(g->d_tricam); g->d_tricam = nullptr; g->tricam_valid = false;
    }
    if (!g->d_tricam) {
        if (hipMalloc((void**)&g->d_tricam, (size_t)g->n * sizeof(TriCam)) != hipSuccess) { (void)hipGetLastError(); g->d_tricam = nullptr; return; }
        g->tricam_n = g->n; g->tricam_valid = false;
    }
    // ... and the planes of the walk's records relative to the camera (same launch, same validity): TLG_AXIS_MAX_HOST records behind the cells'
    if (g->walk_records) {
        t.axis_cam = reinterpret_cast<TlgCamPlane*>(g->d_toprec + g->info.cells);
        t.cam_off = g->info.cells * (uint32_t)sizeof(TlgTopRec);
    }
    const float o[3] = { cam->viewTransform[3], cam->viewTransform[7], cam->viewTransform[11] };
    t.tricam = g->d_tricam;
    t.prepare_tricam = !g->tricam_valid || std::memcmp(o, g->tricam_origin, sizeof o) != 0;
    std::memcpy(g->tricam_origin, o, sizeof o);
    g->tricam_valid = true;
}

int rt_tlg_generate_contacts_primary(rt_tlg* g, const rt_camera* cam, void* d_contacts)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!g->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (!d_contacts) return RT_FAIL(ctx, "contacts buffer is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    RT_ENTER(ctx);
    TlgLaunch t = make_tlg_launch(g);
    use_camera_triangles(g, cam, t);
    launch_tlg_trace_primary(ctx->stream, t, *cam, 0, 1, false, (rt_contact*)d_contacts);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_tlg_generate_frame(rt_tlg* g, const rt_camera* cam, void* d_contacts, void* d_rgba)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!g->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (!d_contacts || !d_rgba) return RT_FAIL(ctx, "null buffer");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    RT_ENTER(ctx);
    TlgLaunch t = make_tlg_launch(g);
    use_camera_triangles(g, cam, t);
    launch_tlg_trace_primary(ctx->stream, t, *cam, 0, 1, false, (rt_contact*)d_contacts, (float4*)d_rgba);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_tlg_generate_contacts_primary_tiled(rt_tlg* g, const rt_camera* cam, uint32_t rank, uint32_t nranks, void* d_contacts_compact)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!g->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (!d_contacts_compact) return RT_FAIL(ctx, "contacts buffer is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    RT_ENTER(ctx);
    TlgLaunch t = make_tlg_launch(g);
    use_camera_triangles(g, cam, t);
    launch_tlg_trace_primary(ctx->stream, t, *cam, rank, nranks, true, (rt_contact*)d_contacts_compact);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_tlg_generate_contacts_rays(rt_tlg* g, const void* d_rays, uint32_t ray_count, void* d_contacts)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!g->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (ray_count == 0) return RT_SUCCESS;
    if (!d_rays || !d_contacts) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    launch_tlg_trace_rays(ctx->stream, make_tlg_launch(g), (const rt_ray*)d_rays, ray_count, (rt_contact*)d_contacts);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_tlg_info_get(const rt_tlg* g, rt_tlg_info* out)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    if (!out) return RT_FAIL(g->ctx, "null argument");
    *out = g->info;
    return RT_SUCCESS;
}

uint32_t rt_tlg_rebuilds_without_readback(const rt_tlg* g) { return g ? g->rebuilds_without_readback : 0u; }
int rt_tlg_walk_records(const rt_tlg* g) { return g && g->built && g->walk_records ? 1 : 0; }

int rt_tlg_read(const rt_tlg* g, int what, void* h_dst, uint64_t bytes)
{
    if (!g) return RT_FAIL(nullptr, "tlg is null");
    rt_ctx* ctx = g->ctx;
    if (!h_dst) return RT_FAIL(ctx, "null argument");
    if (what == RT_TLG_GRID_DATA) {
        if (!g->frame_initialised && !g->built) return RT_FAIL(ctx, "initializeFrame has not run");
        if (bytes != sizeof(rt_grid_data)) return RT_FAIL(ctx, "size mismatch");
        std::memcpy(h_dst, &g->grid, sizeof(rt_grid_data));
        return RT_SUCCESS;
    }
    if (!g->built) return RT_FAIL(ctx, "the grid is not built");
    const void* src = nullptr;
    uint64_t want = 0;
    switch (what) {
    case RT_TLG_TOP_CELLS:   src = g->d_top;         want = (uint64_t)g->info.cells * sizeof(rt_top_cell); break;
    case RT_TLG_TOP_PAIRS:   src = g->d_pairs;       want = (uint64_t)g->info.pairs_pow2 * sizeof(uint2); break;
    case RT_TLG_CELL_RANGES: src = g->d_cell_ranges; want = (uint64_t)g->info.cells * sizeof(uint2); break;
    case RT_TLG_LEAF_PAIRS:  src = g->d_leaf_pairs;  want = (uint64_t)g->info.leaf_pairs_pow2 * sizeof(uint2); break;
    case RT_TLG_LEAF_RANGES: src = g->d_leaf_ranges; want = (uint64_t)g->info.leaf_cells_count * sizeof(uint2); break;
    default: return RT_FAIL(ctx, "unknown build product");
    }
    if (bytes != want) return RT_FAIL(ctx, "size mismatch");
    if (want == 0) return RT_SUCCESS;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipMemcpyAsync(h_dst, src, want, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_SUCCESS;
}

float rt_tlg_pow_third(float x) { return tlg_pow_third(x); }
