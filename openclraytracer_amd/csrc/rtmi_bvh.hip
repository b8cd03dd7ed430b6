// rtmi_bvh.hip -- C ABI, the BVH manager's build side: rt_bvh_create / initialize_frame / construct / destroy and the parity
// read-backs (BVHManager.cpp:57-228).  Host-side orchestration only; the kernels are in build_kernels.hip / sort_scan_kernels.hip.
#include "rtmi_internal.h"
#include "rtmi_host.h"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <sstream>

extern "C" {

// ---------------------------------------------------------------------------------------
// BVH manager
// ---------------------------------------------------------------------------------------
static uint32_t morton_buffer_items(uint32_t n)     // BVHManager.cpp:128-131
{
    uint32_t p = 1;
    while ((uint64_t)p * 2 <= n) p *= 2;
    if (p < n) p *= 2;
    return p;
}

int rt_bvh_create(rt_ctx* ctx, const rt_scene* scene, int parity_mode, rt_bvh** out)
{
    if (!ctx || !scene || !out) return RT_FAIL(ctx, "null argument");
    *out = nullptr;
    if (scene->ctx != ctx) return RT_FAIL(ctx, "scene belongs to another context");
    const bool device_arith = (parity_mode & RT_MODE_DEVICE_ARITH) != 0;
    parity_mode &= ~RT_MODE_DEVICE_ARITH;
    if (parity_mode != RT_MODE_FIXED && parity_mode != RT_MODE_LITERAL_ZERO && parity_mode != RT_MODE_FAST) return RT_FAIL(ctx, "unknown parity mode");
    if (device_arith) (void)rt_preload_trace_devarith();
    rt_bvh* b = new (std::nothrow) rt_bvh();
    if (!b) return RT_FAIL(ctx, "out of host memory");
    b->ctx = ctx; b->scene = scene; b->mode = parity_mode; b->device_arith = device_arith;
    *out = b;
    return RT_SUCCESS;
}


static void drop_build_graph(rt_bvh* b)
{
    if (b->build_exec) (void)hipGraphExecDestroy(b->build_exec);
    if (b->build_graph) (void)hipGraphDestroy(b->build_graph);
    b->build_exec = nullptr; b->build_graph = nullptr; b->builds_on_allocation = 0; b->build_graph_failed = false;
}

static void free_build_buffers(rt_bvh* b)
{
    drop_build_graph(b);                   // it holds the addresses freed below
    if (b->d_nodes) (void)hipFree(b->d_nodes);
    if (b->d_pairs) (void)hipFree(b->d_pairs);
    if (b->d_counters) (void)hipFree(b->d_counters);
    if (b->d_trav) (void)hipFree(b->d_trav);
    if (b->d_quad) (void)hipFree(b->d_quad);
    if (b->d_quad_cam) (void)hipFree(b->d_quad_cam);
    b->d_quad = nullptr; b->d_quad_cam = nullptr;
    if (b->d_depth) (void)hipFree(b->d_depth);
    if (b->d_pos) (void)hipFree(b->d_pos);
    if (b->d_trirec) (void)hipFree(b->d_trirec);
    b->d_trirec = nullptr;
    if (b->d_tricam) (void)hipFree(b->d_tricam);
    b->d_tricam = nullptr;
    if (b->d_packet_spill) (void)hipFree(b->d_packet_spill);
    b->d_packet_spill = nullptr; b->packet_spill_entries = 0;
    if (b->d_tree_select) (void)hipFree(b->d_tree_select);
    b->d_tree_select = nullptr; b->have_search_tree = false;
    if (b->s_nodes) (void)hipFree(b->s_nodes);
    if (b->s_pairs) (void)hipFree(b->s_pairs);
    if (b->s_counters) (void)hipFree(b->s_counters);
    if (b->s_pos) (void)hipFree(b->s_pos);
    if (b->d_sort_tmp2) (void)hipFree(b->d_sort_tmp2);
    if (b->d_sort_scratch2) (void)hipFree(b->d_sort_scratch2);
    b->d_sort_tmp2 = nullptr; b->d_sort_scratch2 = nullptr;
    if (b->d_sort_tmp) (void)hipFree(b->d_sort_tmp);
    if (b->d_sort_scratch) (void)hipFree(b->d_sort_scratch);
    b->d_sort_tmp = nullptr; b->d_sort_scratch = nullptr;
    if (b->s_spans) (void)hipFree(b->s_spans);
    if (b->s_span_scratch) (void)hipFree(b->s_span_scratch);
    if (b->d_span_scratch) (void)hipFree(b->d_span_scratch);
    b->s_spans = nullptr; b->s_span_scratch = nullptr; b->d_span_scratch = nullptr;
    if (b->d_spans) (void)hipFree(b->d_spans);
    if (b->d_rq_scratch) (void)hipFree(b->d_rq_scratch);
    b->d_spans = nullptr; b->d_rq_scratch = nullptr;
    b->s_nodes = nullptr; b->s_pairs = nullptr; b->s_counters = nullptr; b->s_pos = nullptr;
    b->d_nodes = nullptr; b->d_pairs = nullptr; b->d_counters = nullptr; b->d_trav = nullptr;
    b->d_depth = nullptr; b->d_pos = nullptr;
}

void rt_bvh_destroy(rt_bvh* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    free_build_buffers(b);
    drop_rays_tuned(b);
    if (b->h_depths) (void)hipHostFree(b->h_depths);
    block_order_free(b->block_order);
    delete b;
}

// Large scenes refit by range queries instead of climbing parent links (build_kernels.hip 4a): the climb is serial in the depth of
// the reference tree, which duplicate Morton codes turn into chains of thousands of levels.  Same node bytes either way; below
// the threshold the single climbing launch is faster than the dozen small launches of the query structure.
// env RTMI_REFIT=rq / climb forces one (tests run both on every fixture).
static bool refit_by_range_queries(const rt_ctx* ctx, uint32_t n, int mode)       // round 2's sparse table: only on request since round 4
{
    if (mode == RT_MODE_LITERAL_ZERO || n < 256u) return false;
    return ctx->refit_choice == 1;
}
// Refit by spans over a segment tree of the sorted leaf boxes (build_kernels.hip 4a'): no walker waits for another, O(n) to build -- every
// scene from 256 triangles on, both hierarchies (below that a climb of a handful of levels is one launch).  RTMI_REFIT=climb / rq restore
// the older paths (tests run all three on every fixture: same node bytes).
static bool refit_by_spans(const rt_ctx* ctx, uint32_t n, int mode)
{
    if (mode == RT_MODE_LITERAL_ZERO || n < 256u) return false;
    return ctx->refit_choice == 0 || ctx->refit_choice == 3;
}

int rt_bvh_initialize_frame(rt_bvh* b)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    RT_ENTER(ctx);
    uint32_t n = b->scene->d.n_tris;
    if (n < 2) return RT_FAIL(ctx, "a BVH needs at least two triangles");
    uint32_t N = morton_buffer_items(n);
    if (n != b->n) {                                  // grow-only in the reference (CLBuffer::resize); exact here
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_build_buffers(b);
        b->n = 0; b->N = 0;                           // stays 0 unless every allocation below succeeds
        // LITERAL_ZERO: no search tree (its boxes are not nested: the spine above the never-built node keeps the init boxes)
        const bool want_search = !ctx->no_search_tree && b->mode != RT_MODE_LITERAL_ZERO;
        const size_t trees = want_search ? 2u : 1u;
        hipError_t e = hipMalloc((void**)&b->d_nodes, (size_t)(2 * n - 1) * sizeof(rt_bvh_node));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_pairs, (size_t)N * sizeof(uint2));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_counters, (size_t)n * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_trav, (size_t)(n - 1) * sizeof(TravNode) * trees);
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_depth, 256);
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_pos, (size_t)n * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_trirec, (size_t)n * sizeof(TriRec));
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_tree_select, 256);
        if (e == hipSuccess && !ctx->no_packet4) e = hipMalloc((void**)&b->d_quad, (size_t)(n - 1) * sizeof(QuadNode) * trees);
        if (e == hipSuccess && !ctx->no_packet4) {
            // the camera-relative copy, and eight more with the near / far planes already in place for one direction octant
            // each (memory for vector instructions: 6 of the 16 per box test), unless that would be several GB
            const size_t one = (size_t)(n - 1) * sizeof(QuadNode) * trees;
            b->quad_cam_variants = (one * 9 <= ((size_t)4 << 30) && !ctx->no_octant_records) ? 9u : 1u;
            e = hipMalloc((void**)&b->d_quad_cam, one * b->quad_cam_variants);
        }
        if (e == hipSuccess && !ctx->no_packet4) e = hipMalloc((void**)&b->d_tricam, (size_t)n * sizeof(TriCam));
        if (e == hipSuccess && want_search) {   // working set of the search-tree build, kept between builds (an allocation costs ~0.1 ms)
            e = hipMalloc((void**)&b->s_nodes, (size_t)(2 * n - 1) * sizeof(rt_bvh_node));
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_pairs, (size_t)N * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_counters, (size_t)n * sizeof(uint32_t) + 256);
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_pos, (size_t)n * sizeof(uint32_t));
        }
        if (e == hipSuccess && (want_search || b->mode == RT_MODE_FAST)) {      // the radix sort's second buffer and histograms
            e = hipMalloc((void**)&b->d_sort_tmp, (size_t)n * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_sort_scratch, radix_sort_scratch_words(n) * sizeof(uint32_t));
        }
        if (e == hipSuccess && want_search && b->mode == RT_MODE_FAST) {        // ... and the reference tree's own: both sorts go through the same launches
            e = hipMalloc((void**)&b->d_sort_tmp2, (size_t)n * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void**)&b->d_sort_scratch2, radix_sort_scratch_words(n) * sizeof(uint32_t));
        }
        if (e == hipSuccess && refit_by_range_queries(ctx, n, b->mode)) {
            e = hipMalloc((void**)&b->d_spans, (size_t)(n - 1) * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc(&b->d_rq_scratch, refit_rq_scratch_bytes(n));
        }
        if (e == hipSuccess && refit_by_spans(ctx, n, b->mode)) {
            e = hipMalloc((void**)&b->d_spans, (size_t)(n - 1) * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc(&b->d_span_scratch, refit_span_scratch_bytes(n));
            if (e == hipSuccess) e = hipMemsetAsync(b->d_span_scratch, 0, refit_span_scratch_bytes(n), ctx->stream);       // (its ticket counter starts at zero and is left at zero by every build)
            if (e == hipSuccess && want_search) e = hipMalloc((void**)&b->s_spans, (size_t)(n - 1) * sizeof(uint2));
            if (e == hipSuccess && want_search) e = hipMalloc(&b->s_span_scratch, refit_span_scratch_bytes(n));
            if (e == hipSuccess && want_search) e = hipMemsetAsync(b->s_span_scratch, 0, refit_span_scratch_bytes(n), ctx->stream);
        }
        if (e != hipSuccess) {
            free_build_buffers(b);
            (void)hipGetLastError();
            return RT_FAIL(ctx, std::string("initializeFrame: device allocation failed: ") + hipGetErrorString(e));
        }
        b->n = n; b->N = N;
        b->build_generation++; b->eager_frames = 0;
    }
    // BVHManager.cpp:134-147: morton buffer := (UINT_MAX,UINT_MAX), visit counters := 0.
    // The node buffer is zeroed as well: FIXED mode is defined on zero-initialised buffers.
    // (enqueued at the head of construct(): one stream-ordered sequence, which a rebuild replays as one graph)
    b->clears_pending = true;
    if (!b->h_depths) RT_HIP(ctx, hipHostMalloc((void**)&b->h_depths, 64));
    b->frame_initialised = true;
    b->built = false;
    return RT_SUCCESS;
}

// everything a build enqueues, in stream order and nothing else (no allocation, no synchronisation: it is also what a capture records)
static int enqueue_build(rt_bvh* b, hipStream_t s)
{
    rt_ctx* ctx = b->ctx;
    const DeviceScene& sc = b->scene->d;
    const uint32_t n = b->n, N = b->N;
    // BVHManager.cpp:134-147 (initializeFrame): morton buffer := (UINT_MAX,UINT_MAX), visit counters := 0.  A build is a chain of dependent
    // launches of a few microseconds each (rocprofv3: every launch of a 50 032-triangle build takes 4.5-6 us, the clearing fills included), so
    //  * a clear that a kernel of the chain can do on its way is not launched: the pair buffers' padding is written by k_morton_leaves, the
    //    counters are cleared only where a climb will use them, and the node buffer -- zeroed in the reference's compatibility mode, whose
    //    never-built node must read as zeros -- is written whole by FIXED / FAST builds (leaves by k_morton_leaves, every field of all n - 1
    //    inner nodes by k_radix_tree, each node's parent link by its parent);
    //  * the two hierarchies -- the reference tree and the search tree for primary rays (same leaves, better order) -- go
    //    through the SAME launches: every kernel of the chain takes a second set of buffers and works on it in the blocks with blockIdx.y = 1.
    //    (Two chains side by side on two queues of a graph were tried first: a dependency between two queues costs ~35 us where it is waited
    //    for, the fork and the join together about what the overlap gained.)
    const bool literal = b->mode == RT_MODE_LITERAL_ZERO;
    const bool nan_possible = b->scene->has_nan;
    const bool want_search = !ctx->no_search_tree && !literal;
    const bool fast = b->mode == RT_MODE_FAST;
    // 2. the sort (BVHManager.cpp:195).  The reference's network for the reference tree (the order of equal Morton codes is part of its
    // result: it decides the radix tree's topology and the depth-first ranks that break ties between equal t); RT_MODE_FAST: a stable radix
    // sort instead (30-bit codes: four passes over n pairs instead of ~log^2 N / 2 network levels over N) -- the same tree wherever codes
    // differ, leaves with EQUAL codes in index order.  The search tree's order is the library's own (any nested hierarchy over the reference's
    // leaves gives the reference's contacts, ties between equal t are broken by the REFERENCE rank, d_pos): its keys ride along in whatever
    // sorts the reference's -- the network up to 2^18 pairs, where a launch costs more than what it moves -- or take the radix sort.
    const bool search_with_network = want_search && !fast && (N <= (1u << 18) || !b->d_sort_tmp);
    const bool search_with_radix_pair = want_search && fast && b->d_sort_tmp2;
    const bool ref_climbs = literal || !b->d_spans || nan_possible;
    const bool search_spans = b->s_spans && b->s_span_scratch;
    const bool search_climbs = want_search && (!search_spans || nan_possible);
    uint32_t* const d_depth = b->d_depth;                     // [0] depth, [1] NaN flag of the span refit; [2], [3] the same of the search tree
    uint32_t* const s_depth = b->d_depth + 2;
    if (literal) RT_HIP(ctx, hipMemsetAsync(b->d_nodes, 0, (size_t)(2 * n - 1) * sizeof(rt_bvh_node), s));
    // 1. BVHManager.cpp:183: leaves and Morton pairs (+ the padding of the pair buffer; + the search tree's copy of the leaves and its keys --
    // Morton codes of the box centres normalised to the scene header's box -- and the padding its sort wants)
    // (+ the arrival counters of the climbs, where a climb will run, zeroed on the way)
    if (want_search) launch_morton_leaves(s, sc, b->d_nodes, b->d_pairs, N, b->s_nodes, b->s_pairs, search_with_network ? N : 0u,
                                          ref_climbs ? b->d_counters : nullptr, search_climbs ? b->s_counters : nullptr, !ctx->search_keys_ten_per_axis);
    else launch_morton_leaves(s, sc, b->d_nodes, b->d_pairs, N, nullptr, nullptr, 0u, ref_climbs ? b->d_counters : nullptr, nullptr);
    if (fast) {
        if (search_with_radix_pair) launch_radix_sort(s, b->d_pairs, b->d_sort_tmp2, n, 30u, b->d_sort_scratch2, b->s_pairs, b->d_sort_tmp, b->d_sort_scratch);
        else {
            launch_radix_sort(s, b->d_pairs, b->d_sort_tmp, n, 30u, b->d_sort_scratch);
            // (no second set of sort buffers: the search tree's keys go through the same ones, behind the reference's -- never left unsorted)
            if (want_search) launch_radix_sort(s, b->s_pairs, b->d_sort_tmp, n, 30u, b->d_sort_scratch);
        }
    } else {
        launch_bitonic_sort(s, b->d_pairs, N, search_with_network ? b->s_pairs : nullptr);
        if (want_search && !search_with_network) launch_radix_sort(s, b->s_pairs, b->d_sort_tmp, n, 30u, b->d_sort_scratch);
    }
    // 3. :204 (LITERAL_ZERO: the guard as written, BVHKernels.cl:58); clears the depth + NaN flag words of the refit behind it
    // (the search tree's inner nodes need no clearing either: k_radix_tree writes every field of all n - 1 of them, and that tree is never
    //  read back as "the reference on zeroed buffers")
    if (want_search) launch_radix_tree(s, b->d_nodes, b->d_pairs, n, n - 1u, b->d_spans, d_depth, b->s_nodes, b->s_pairs, b->s_spans, s_depth);
    else launch_radix_tree(s, b->d_nodes, b->d_pairs, n, literal ? n - 2u : n - 1u, b->d_spans, d_depth);
    // 4. :217 the boxes (+ the tree depths).  By spans (every scene from 256 triangles on; or, on request, round 2's range queries); NaN boxes
    // (not associative) send it back to the climb -- decided on the device: the climb is launched behind it and runs only if the pass raised
    // its flag, and only for scenes in which the upload saw a NaN vertex.  The search tree's boxes are plain unions (`tight`).
    const bool tight = !ctx->search_tree_reference_boxes;
    const bool ref_by_spans = !literal && b->d_spans && b->d_span_scratch;
    const bool search_by_spans = want_search && search_spans;
    if (literal) launch_refit_literal(s, b->d_nodes, b->d_counters, n, d_depth);
    else if (ref_by_spans && search_by_spans) {
        launch_refit_span(s, b->d_nodes, b->d_pairs, b->d_spans, n, b->d_span_scratch, d_depth, d_depth + 1, false,
                          b->s_nodes, b->s_pairs, b->s_spans, b->s_span_scratch, s_depth, s_depth + 1, tight, true);
        if (nan_possible) launch_refit_pair(s, n, b->d_nodes, b->d_counters, d_depth, d_depth + 1, b->s_nodes, b->s_counters, s_depth, s_depth + 1, tight);
    }
    else if (!ref_by_spans && !search_by_spans && want_search && !(b->d_spans && b->d_rq_scratch))
        launch_refit_pair(s, n, b->d_nodes, b->d_counters, d_depth, nullptr, b->s_nodes, b->s_counters, s_depth, nullptr, tight);
    else {          // one hierarchy, or the two refitted differently (RTMI_REFIT=rq: the reference tree by range queries)
        if (ref_by_spans) {
            launch_refit_span(s, b->d_nodes, b->d_pairs, b->d_spans, n, b->d_span_scratch, d_depth, d_depth + 1, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, true);
            if (nan_possible) launch_refit(s, b->d_nodes, b->d_counters, n, d_depth, false, d_depth + 1);
        } else if (b->d_spans && b->d_rq_scratch) {
            launch_refit_rq(s, b->d_nodes, b->d_pairs, b->d_spans, n, b->d_rq_scratch, d_depth, d_depth + 1);
            if (nan_possible) launch_refit(s, b->d_nodes, b->d_counters, n, d_depth, false, d_depth + 1);
        } else launch_refit(s, b->d_nodes, b->d_counters, n, d_depth);
        if (search_by_spans) {
            launch_refit_span(s, b->s_nodes, b->s_pairs, b->s_spans, n, b->s_span_scratch, s_depth, s_depth + 1, tight, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, true);
            if (nan_possible) launch_refit(s, b->s_nodes, b->s_counters, n, s_depth, tight, s_depth + 1);
        } else if (want_search) launch_refit(s, b->s_nodes, b->s_counters, n, s_depth, tight);
    }
    // 5. the end of the chain in one launch (build_kernels.hip 7a): traversal records, depth-first ranks, packed triangle records (the search tree's
    // records behind the reference tree's: TravNode[2(n-1)]); the four-wide records -- of the search tree only: no kernel walks the reference tree
    // four records wide (the policy's condition and `cur >= search_root` in the kernels), so that half of the buffer stays unwritten --; the tree
    // depths out of the span refit's marks; and the depth / flag words handed to the host's pinned copy.
    void* const marks_ref = (ref_by_spans ? b->d_span_scratch : nullptr);
    void* const marks_search = (search_by_spans ? b->s_span_scratch : nullptr);
    if (want_search) launch_finish(s, n, ctx->quad_by_area, sc, b->d_trirec, b->d_nodes, b->d_pairs, b->d_trav, b->d_pos, 0u, nullptr, marks_ref, d_depth, b->h_depths,
                                   b->s_nodes, b->s_pairs, b->d_trav + (n - 1), b->s_pos, n - 1u, b->d_quad ? b->d_quad + (n - 1) : nullptr, marks_search, s_depth, b->h_depths + 2);
    else launch_finish(s, n, ctx->quad_by_area, sc, b->d_trirec, b->d_nodes, b->d_pairs, b->d_trav, b->d_pos, 0u, literal ? nullptr : b->d_quad, marks_ref, d_depth, b->h_depths);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_bvh_construct(rt_bvh* b)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->frame_initialised) return RT_FAIL(ctx, "initializeFrame must precede construct");
    RT_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const bool literal = b->mode == RT_MODE_LITERAL_ZERO;
    const bool want_search = !ctx->no_search_tree && !literal;
    b->have_search_tree = false;
    for (int k = 0; k < 4; k++) b->h_depths[k] = 0u;
    // A scene that moves is rebuilt every frame (OpenCLRayTracerDemo.cpp:230-234): the same ~40 launches, clears and copies on the same
    // buffers.  The first build of an allocation is enqueued as it is, the second is captured into a graph, and from then on a build is
    // one graph launch (12 triangles: 0.108 -> see DESIGN.md; the host issued the launches more slowly than the device ran them).
    // RTMI_BUILD_GRAPH=0: always eager.
    const bool graphs = !ctx->no_build_graph;          // knob RTMI_BUILD_GRAPH=0
    b->builds_on_allocation++;
    bool launched = false;
    if (graphs && b->builds_on_allocation >= 2u) {
        if (!b->build_exec && !b->build_graph_failed) {
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            if (e == hipSuccess) {
                const int rc = enqueue_build(b, s);
                e = hipStreamEndCapture(s, &g);
                if (rc == RT_SUCCESS && e == hipSuccess && g && hipGraphInstantiate(&b->build_exec, g, nullptr, nullptr, 0) == hipSuccess) b->build_graph = g;
                else { if (g) (void)hipGraphDestroy(g); b->build_exec = nullptr; }
            }
            if (!b->build_exec) b->build_graph_failed = true;      // not tried again on this allocation (drop_build_graph resets it)
            (void)hipGetLastError();                   // a runtime that cannot capture this sequence: the build stays eager
        }
        if (b->build_exec) { RT_HIP(ctx, hipGraphLaunch(b->build_exec, s)); launched = true; b->graph_builds++; }
    }
    if (!launched && enqueue_build(b, s)) return RT_ERROR;
    b->clears_pending = false;
    RT_HIP(ctx, hipStreamSynchronize(s));                          // construct() returns with the build finished
    const uint32_t depth = b->h_depths[0], search_depth = want_search ? b->h_depths[2] : 0u;
    b->max_depth = depth;
    b->search_depth = search_depth;
    // (a degenerate model -- thousands of coincident box centres -- makes this tree as deep as the reference's: no gain then,
    // but the four-wide records may only be used on the search tree, so it is used whenever it exists)
    b->have_search_tree = want_search && search_depth != 0;
    {   // packet kernel: (node, lane mask) stack entries beyond the LDS part live in a per-wave global area
        // the four-wide packet kernel pushes up to three entries per two levels
        // (the packet kernels walk the search tree by default, whose depth is independent of the reference tree's)
        const uint32_t deeper = depth > search_depth ? depth : search_depth;
        const uint32_t deepest = b->d_quad ? 2u * deeper + 16u : deeper + 8u;
        uint32_t need = deepest > trace_packet_stack_entries() ? deepest - trace_packet_stack_entries() : 0u;
        if (need != b->packet_spill_entries) {
            if (b->d_packet_spill) { (void)hipFree(b->d_packet_spill); b->d_packet_spill = nullptr; }
            b->packet_spill_entries = need;
            // one area per wave the device can hold at once (a wave takes one only when its LDS stack overflows)
            b->packet_spill_slots = ((uint32_t)ctx->prop.multiProcessorCount * (uint32_t)(ctx->prop.maxThreadsPerMultiProcessor / 64) + 31u) & ~31u;
            if (need) {
                const size_t bytes = ((size_t)(b->packet_spill_slots >> 5) + (size_t)b->packet_spill_slots * need * 3u) * sizeof(uint32_t);
                RT_HIP(ctx, hipMalloc((void**)&b->d_packet_spill, bytes));
                RT_HIP(ctx, hipMemsetAsync(b->d_packet_spill, 0, (size_t)(b->packet_spill_slots >> 5) * sizeof(uint32_t), s));   // all areas free
                RT_HIP(ctx, hipStreamSynchronize(s));
            }
        }
    }
    b->frame_initialised = false;
    b->built = true;
    b->marked_valid = false;
    b->build_generation++; b->eager_frames = 0;            // (the host's copy of the planes is this build's only after a few frames on it)
    // Dynamic scenes rebuild every frame (OpenCLRayTracerDemo.cpp:230-234).  The calibrated strategy is a property of how the
    // model fills the frame, not of one build: it is kept while the triangle count stays (the key holds resolution and tiling),
    // and every strategy produces the same contacts anyway.  RTMI_RECALIBRATE=1 restores a calibration per construct().
    if (b->tuned_n != b->n || ctx->recalibrate) { b->tuned = false; for (auto& e : b->tuned_cache) e.valid = false; }
    b->tuned_n = b->n;
    if (b->rays_tuned_n != b->n || ctx->recalibrate) drop_rays_tuned(b);     // (the ray-buffer entry's strategies likewise)
    b->rays_tuned_n = b->n;
    return RT_SUCCESS;
}

uint32_t rt_bvh_graph_builds(const rt_bvh* b) { return b ? b->graph_builds : 0u; }

int rt_bvh_read_nodes(const rt_bvh* b, void* h_nodes)
{
    if (!b || !h_nodes) return RT_FAIL(b ? b->ctx : nullptr, "null argument");
    if (!b->built) return RT_FAIL(b->ctx, "BVH not built");
    return rt_buf_download(b->ctx, h_nodes, b->d_nodes, (uint64_t)(2 * b->n - 1) * sizeof(rt_bvh_node));
}

int rt_bvh_read_search_nodes(const rt_bvh* b, void* h_nodes)
{
    if (!b || !h_nodes) return RT_FAIL(b ? b->ctx : nullptr, "null argument");
    if (!b->built) return RT_FAIL(b->ctx, "BVH not built");
    if (!b->have_search_tree || !b->s_nodes) return RT_FAIL(b->ctx, "this BVH has no search tree");
    return rt_buf_download(b->ctx, h_nodes, b->s_nodes, (uint64_t)(2 * b->n - 1) * sizeof(rt_bvh_node));
}

int rt_bvh_read_morton_pairs(const rt_bvh* b, void* h_pairs, uint32_t* n_pow2)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    if (!b->built) return RT_FAIL(b->ctx, "BVH not built");
    if (n_pow2) *n_pow2 = b->N;
    if (!h_pairs) return RT_SUCCESS;
    return rt_buf_download(b->ctx, h_pairs, b->d_pairs, (uint64_t)b->N * sizeof(uint2));
}

int rt_bvh_info(const rt_bvh* b, uint32_t* leaves, uint32_t* morton_items, uint32_t* max_depth)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    if (leaves) *leaves = b->n;
    if (morton_items) *morton_items = b->N;
    if (max_depth) *max_depth = b->max_depth;
    return RT_SUCCESS;
}

}  // extern "C"
