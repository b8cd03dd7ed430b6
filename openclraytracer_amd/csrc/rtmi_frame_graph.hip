// rtmi_frame_graph.hip -- C ABI, frame graphs: the launch sequence of one steady-state frame captured into a HIP graph, replayed with one
// call, and followed by a camera that moves (rt_frame_graph_*; the Demo's RTFrame loop, OpenCLRayTracerDemo.cpp:271-326).
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
// frame graphs: the steady-state launch sequence of one frame (control-block memset, first-phase kernel, wave-per-ray
// kernel, shading) captured once into a HIP graph and replayed with one call.  The kernels of an easy frame run for
// 60-100 us while issuing them takes 30-40 us of host time, and with the frame tiled over N GPUs the host becomes the
// bound.
// ---------------------------------------------------------------------------------------
struct rt_frame_graph {
    rt_ctx* ctx = nullptr;
    rt_bvh* bvh = nullptr;
    const rt_scene* scene = nullptr;
    // the frame: everything frame_issue needs to issue it again (eagerly, or as new arguments of the captured kernels)
    rt_camera cam{};
    uint32_t rank = 0, nranks = 1, slots_per_rank = 0;
    int layout = RT_FRAME_UNTILED;
    void* d_contacts = nullptr; void* d_out = nullptr;
    // [0] the steady-state frame; [1] the same behind the per-camera preparation (k_prepare_camera): what a frame costs whose
    // camera has moved.  Moving the camera re-sets the arguments of the captured kernels, it does not capture again.
    // Two executable instances of each: the arguments of an instance are only re-set once its last launch has finished (an event
    // behind every launch), so a frame in flight is never touched -- with two, the host prepares the next camera's frame while
    // the GPU still runs the previous one.
    hipGraph_t graph[2] = { nullptr, nullptr };
    hipGraphExec_t exec[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
    hipEvent_t done[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
    bool launched[2][2] = { { false, false }, { false, false } };
    bool has_event[2][2] = { { false, false }, { false, false } };   // a `done` event stands behind the instance's last launch
    bool nodes_known = true;                 // the kernel nodes of both captured sequences were identified (else a moved camera is rendered eagerly)
    bool moved = false;                      // rt_frame_graph_set_camera was called: launches leave events behind (a graph whose camera never moves pays nothing)
    bool dirty[2][2] = { { false, false }, { false, false } };   // the camera changed since this instance's arguments were set
    int slot[2] = { 0, 0 };                  // the instance of each kind that was launched last
    GraphNodes nodes[2];
    // what the captured launches depend on besides the caller's buffers: addresses of the context's scratch lists and of the
    // BVH's records (generations), and the per-camera records of this camera position (origin, parts)
    uint64_t scratch_generation = 0, build_generation = 0;
    float origin[3] = { 0, 0, 0 };
    uint32_t parts = 0;                      // RT_PREP_* this frame's kernels read
    bool packet = false, quad = false;       // the strategy the frame was captured with (kept whatever the BVH is calibrated to later)
    uint32_t budget = 0;
    bool four_wide = false, octant_records = false;   // ... the four-wide packet kernel (on per-octant records: x-plane rays stay)
    bool no_unsafe = false;                  // captured without the wave-per-ray launch (and, unless switched off, with fused shading)
    bool eager = false;                      // the current camera cannot be served by the captured sequence: launch issues the frame eagerly
    uint32_t plane_bits_known = 0;
    uint32_t waves_per_block = 0;            // workgroup size of the first-phase kernel, as captured
    bool direct[2] = { false, false };       // the sequence holds kernel nodes only and is issued as plain launches (RTMI_GRAPH_DIRECT=1)
    BlockOrder bo;                           // this frame's start order of the blocks (its buffers' addresses are arguments of the captured kernel)
    bool bo_sorted = false;
    uint32_t bo_interval = RT_BLOCK_ORDER_REFRESH;   // frames until the next sort of this graph's order
    int last_launch = -1;                    // 0 steady state, 1 with the preparation, 2 eager
    double host_update_us = 0.0, host_launch_us = 0.0, host_set_camera_us = 0.0;   // RTMI_GRAPH_DEBUG=1: host time per stage
    uint64_t host_updates = 0, host_launches = 0, host_set_cameras = 0;
};

static int frame_issue(rt_bvh* b, const rt_scene* scene, const rt_camera* cam, uint32_t rank, uint32_t nranks, int layout,
                       void* d_contacts, void* d_out, uint32_t slots_per_rank, FrameMode* mode = nullptr)
{
    rt_ctx* ctx = b->ctx;
    const bool tiled = layout != RT_FRAME_UNTILED;
    // a captured frame whose camera sits on no box plane lets the four-wide kernel shade what it stores (trace_kernels.hip, fused
    // frame); whether it did is its launcher's answer -- any other strategy leaves the colours to k_shade
    FusedShade fused{ d_out, layout == RT_FRAME_TILED_PACKED ? slots_per_rank : 0u, false };
    FrameMode plain;
    FrameMode* m = mode ? mode : &plain;
    // (... and a captured frame of the lane-per-ray strategy lets its two kernels shade what they store)
    // (RT_MODE_DEVICE_ARITH: the trace kernels' dot() is the fused one; shading is host code in the reference and stays with k_shade)
    m->fused = (b->capture_no_unsafe_rays || (m->fixed_strategy && !m->packet) || m->eager_fused) && !ctx->no_fused_frames && !b->device_arith ? &fused : nullptr;
    const bool update = m->update != nullptr;
    if (trace_primary_tuned(b, cam, tiled ? rank : 0u, tiled ? nranks : 1u, tiled, (rt_contact*)d_contacts, m)) return RT_ERROR;
    if (fused.done) { if (!update) RT_HIP(ctx, hipGetLastError()); return RT_SUCCESS; }
    if (!tiled)
        launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_contacts, cam->resX * cam->resY, (float4*)d_out, true, 0, 1, 0, m->record, m->update);
    else
        launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_contacts, rt_tiles_owned_pixels(cam, rank, nranks), (float4*)d_out,
                     true, rank, nranks, layout == RT_FRAME_TILED_PACKED ? slots_per_rank : 0u, m->record, m->update);
    if (update) return m->update->failed ? RT_FAIL(ctx, "frame graph: a kernel node could not be updated") : RT_SUCCESS;
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

// One frame, eagerly: generateContacts + shade as the Demo's RTFrame does (OpenCLRayTracerDemo.cpp:287-319: contacts, then colours), issued as
// ONE sequence in which the trace kernels shade what they store where the strategy allows it (the four-wide and the lane-per-ray kernels with
// the wave-per-ray kernel behind them; otherwise k_shade follows).  What a scene that is rebuilt every frame has instead of a frame graph (a
// graph does not survive construct()): the same contacts as rt_bvh_generate_contacts_primary[_tiled], the same colours as
// rt_shade_blinn_phong_frame / _tiled / _tiled_packed within the 1e-5 of the fused path.
int rt_render_frame(rt_bvh* b, const rt_scene* scene, const rt_camera* cam, uint32_t rank, uint32_t nranks, int layout,
                    void* d_contacts, void* d_out, uint32_t slots_per_rank)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!scene || !d_contacts || !d_out) return RT_FAIL(ctx, "null argument");
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (layout != RT_FRAME_UNTILED && layout != RT_FRAME_TILED && layout != RT_FRAME_TILED_PACKED) return RT_FAIL(ctx, "unknown frame layout");
    if (layout == RT_FRAME_UNTILED) { rank = 0; nranks = 1; }
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    if (layout == RT_FRAME_TILED_PACKED && slots_per_rank < rt_tiles_owned_pixels(cam, rank, nranks)) return RT_FAIL(ctx, "slots_per_rank too small");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, (uint64_t)rt_tiles_owned_pixels(cam, rank, nranks))) return RT_ERROR;
    FrameMode m;
    m.eager_fused = !ctx->profiling;          // (per-launch profiling times the trace kernels alone)
    return frame_issue(b, scene, cam, rank, nranks, layout, d_contacts, d_out, slots_per_rank, &m);
}

static FrameMode graph_mode(rt_frame_graph* g, int which, bool update)
{
    FrameMode m;
    if (update) m.update = &g->nodes[which]; else m.record = &g->nodes[which];
    m.force_prepare = which == 1;
    m.plane_bits_known = g->plane_bits_known;
    // without the wave-per-ray kernel nothing reads the marks of the binary records, and the plane bits come by value
    m.parts_drop = (g->four_wide && g->no_unsafe) ? RT_PREP_MARKS : 0u;
    m.fixed_strategy = true; m.packet = g->packet; m.quad = g->quad; m.budget = g->budget; m.waves_per_block = g->waves_per_block;
    m.block_order = &g->bo;
    return m;
}

static int graph_issue(rt_frame_graph* g, FrameMode* m)
{
    rt_bvh* b = g->bvh;
    b->capture_no_unsafe_rays = g->no_unsafe;
    const int rc = frame_issue(b, g->scene, &g->cam, g->rank, g->nranks, g->layout, g->d_contacts, g->d_out, g->slots_per_rank, m);
    b->capture_no_unsafe_rays = false;
    return rc;
}

static int capture_frame(rt_frame_graph* g, int which)
{
    rt_ctx* ctx = g->ctx;
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return RT_FAIL(ctx, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    FrameMode m = graph_mode(g, which, false);
    const int rc = graph_issue(g, &m);
    e = hipStreamEndCapture(ctx->stream, &g->graph[which]);
    if (rc != RT_SUCCESS || e != hipSuccess || !g->graph[which]) {
        (void)hipGetLastError();
        return rc != RT_SUCCESS ? RT_ERROR : RT_FAIL(ctx, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    // (a runtime that does not report the captured nodes -- hipStreamGetCaptureInfo_v2 -- still replays the frame; only a camera that
    //  moves is then served by the eager sequence, rt_frame_graph_set_camera)
    if (g->nodes[which].failed) g->nodes_known = false;
    {
        // a sequence of kernels and nothing else (no memset, no copy) is issued as plain launches of the recorded kernels with the
        // recorded arguments: on one stream they follow each other with a shorter gap than graph launches do (a scene's frames back
        // to back: ManFigure 0.127 -> 0.090 ms per frame, cube 0.058 -> 0.049; five scenes' moving cameras 0.272 -> 0.263 ms per step).
        // RTMI_GRAPH_DIRECT=0 keeps hipGraphLaunch.
        const bool want_direct = !ctx->no_graph_direct;
        size_t n_nodes = 0, n_kernels = 0;
        for (int kind = 0; kind < RT_NODE_KINDS; kind++) if (g->nodes[which].node[kind]) n_kernels++;
        g->direct[which] = want_direct && !g->nodes[which].failed && hipGraphGetNodes(g->graph[which], nullptr, &n_nodes) == hipSuccess && n_nodes == n_kernels && n_kernels != 0;
    }
    for (int k = 0; k < 2; k++) {
        e = hipGraphInstantiate(&g->exec[which][k], g->graph[which], nullptr, nullptr, 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done[which][k], hipEventDisableTiming);
        if (e != hipSuccess) return RT_FAIL(ctx, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    return RT_SUCCESS;
}

int rt_frame_graph_create(rt_bvh* b, const rt_scene* scene, const rt_camera* cam, uint32_t rank, uint32_t nranks, int layout,
                          void* d_contacts, void* d_out, uint32_t slots_per_rank, rt_frame_graph** out)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!out) return RT_FAIL(ctx, "out is null");
    *out = nullptr;
    if (!scene || !d_contacts || !d_out) return RT_FAIL(ctx, "null argument");
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (layout != RT_FRAME_UNTILED && layout != RT_FRAME_TILED && layout != RT_FRAME_TILED_PACKED) return RT_FAIL(ctx, "unknown frame layout");
    if (layout == RT_FRAME_UNTILED) { rank = 0; nranks = 1; }
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    if (layout == RT_FRAME_TILED_PACKED && slots_per_rank < rt_tiles_owned_pixels(cam, rank, nranks)) return RT_FAIL(ctx, "slots_per_rank too small");
    if (ctx->profiling) return RT_FAIL(ctx, "frame graphs cannot be captured while per-launch profiling is on");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, (uint64_t)rt_tiles_owned_pixels(cam, rank, nranks))) return RT_ERROR;
    // one eager frame first: it calibrates the strategy for this (resolution, tiling), prepares the camera and sizes every
    // buffer, none of which may happen during a capture
    if (frame_issue(b, scene, cam, rank, nranks, layout, d_contacts, d_out, slots_per_rank)) return RT_ERROR;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (check_trace_error(ctx)) return RT_ERROR;
    rt_frame_graph* g = new (std::nothrow) rt_frame_graph();
    if (!g) return RT_FAIL(ctx, "out of host memory");
    g->ctx = ctx; g->bvh = b; g->scene = scene;
    g->cam = *cam; g->rank = rank; g->nranks = nranks; g->layout = layout; g->slots_per_rank = slots_per_rank;
    g->d_contacts = d_contacts; g->d_out = d_out;
    g->scratch_generation = ctx->scratch_generation; g->build_generation = b->build_generation;
    g->origin[0] = cam->viewTransform[3]; g->origin[1] = cam->viewTransform[7]; g->origin[2] = cam->viewTransform[11];
    bool packet = false, quad = false;
    current_strategy(b, packet, quad);
    g->packet = packet; g->quad = quad;
    g->waves_per_block = ctx->trace_workgroup_waves ? ctx->trace_workgroup_waves : (ctx->frames_in_flight > 1u ? 1u : 0u);
    g->budget = (ctx->defer_budget == 0 && b->tuned) ? b->tuned_budget : ctx->defer_budget;
    g->four_wide = packet && quad;
    g->octant_records = g->four_wide && b->quad_cam_variants == 9u;
    // The four-wide packet kernel hands rays to the wave-per-ray kernel only when the camera sits on a box plane (ray_root; on its
    // per-octant records: on a y or z plane): the eager frame above has computed those bits for this camera position on the device -- unless
    // the host already knows the planes of this build and looked the position up itself (then the host's answer is taken here too) -- so a
    // captured frame of a camera on no such plane leaves the (then empty, ~7 us) wave-per-ray launch out.  When the camera
    // moves (rt_frame_graph_set_camera) the bits are looked up on the host.
    uint32_t plane_bits = 7u;
    if (b->have_search_tree && b->h_planes_generation == b->build_generation && !b->h_planes[0].empty()) {
        // the host knows this build's planes (eager frames beyond RT_EAGER_PLANES_AFTER, or a moved graph): the eager frame above may have
        // left the per-camera marks out on the strength of them, in which case the device word belongs to an earlier camera position
        plane_bits = host_plane_bits(b, g->origin);
    } else if (b->d_tree_select && b->have_search_tree) {
        const hipError_t e = hipMemcpy(&plane_bits, b->d_tree_select, sizeof plane_bits, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete g; return RT_FAIL(ctx, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    }
    g->plane_bits_known = 0x80000000u | (plane_bits & 7u);
    g->no_unsafe = g->four_wide && (plane_bits & (g->octant_records ? 6u : 7u)) == 0u;
    g->parts = strategy_parts(b, cam, packet, quad) & ~((g->four_wide && g->no_unsafe) ? RT_PREP_MARKS : 0u);
    if (g->four_wide && !ctx->no_block_order && (ctx->frames_in_flight <= 1u || ctx->order_in_flight) && block_order_fit(ctx, g->bo, (rt_tiles_owned_pixels(cam, rank, nranks) + 63u) / 64u)) {
        // the captured kernel reads its start order from this graph's own buffer: index order until the first launches have been counted
        g->bo.valid = launch_block_order(ctx->stream, g->bo.d_cost, g->bo.d_order, g->bo.count, ctx->block_order_group_log, true);
        if (!g->bo.valid) block_order_free(g->bo);
    }
    for (int which = 0; which < 2; which++) {
        if (capture_frame(g, which) != RT_SUCCESS) { rt_frame_graph_destroy(g); return RT_ERROR; }
    }
    *out = g;
    return RT_SUCCESS;
}

// The Demo moves its camera between frames (OpenCLRayTracerDemo.cpp:275-276 sets position and orientation before RTFrame,
// key / mouse handlers :378-460): the captured kernels get the new camera as new arguments, and the next launch replays the
// sequence that starts with the per-camera preparation.  Same resolution, buffers and tiling as at creation.
int rt_frame_graph_set_camera(rt_frame_graph* g, const rt_camera* cam)
{
    if (!g) return RT_FAIL(nullptr, "frame graph is null");
    rt_ctx* ctx = g->ctx;
    rt_bvh* b = g->bvh;
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (cam->resX != g->cam.resX || cam->resY != g->cam.resY) return RT_FAIL(ctx, "frame graph: the resolution is part of the graph; create a new one");
    if (g->build_generation != b->build_generation || !b->built)
        return RT_FAIL(ctx, "frame graph is stale: the BVH was rebuilt since the graph was created; create a new graph");
    if (std::memcmp(cam, &g->cam, sizeof *cam) == 0) return RT_SUCCESS;
    RT_ENTER(ctx);
    g->cam = *cam;
    g->origin[0] = cam->viewTransform[3]; g->origin[1] = cam->viewTransform[7]; g->origin[2] = cam->viewTransform[11];
    for (int a = 0; a < 2; a++) for (int k = 0; k < 2; k++) g->dirty[a][k] = true;
    g->eager = !g->nodes_known;
    // a sequence with memset / copy nodes is replayed through hipGraphLaunch and would need its kernel nodes' arguments re-set: not under an
    // older serving runtime (rt_ctx_create: three times slower replays, stalls) -- the eager calls render a moved camera there
    if (ctx->older_runtime && !(g->direct[0] && g->direct[1])) g->eager = true;
    g->moved = true;
    g->bo_interval = RT_BLOCK_ORDER_REFRESH;
    if (g->four_wide) {
        if (ensure_host_planes(b)) return RT_ERROR;
        const uint32_t bits = host_plane_bits(b, g->origin);
        g->plane_bits_known = 0x80000000u | bits;
        // a frame captured without the wave-per-ray kernel cannot walk rays that must use the reference tree: such a camera
        // position (measure zero, but "x = 0" happens) gets the eager sequence until the camera leaves the plane
        if (g->no_unsafe && (bits & (g->octant_records ? 6u : 7u)) != 0u) g->eager = true;
        if (!g->nodes_known) g->eager = true;
    } else {
        g->plane_bits_known = 0;              // the preparation node computes the bits on the device, the kernels read them there
    }
    g->parts = strategy_parts(b, cam, g->packet, g->quad) & ~((g->four_wide && g->no_unsafe) ? RT_PREP_MARKS : 0u);
    return RT_SUCCESS;
}

int rt_frame_graph_last_launch(const rt_frame_graph* g) { return g ? g->last_launch : -1; }

int rt_frame_graph_state(const rt_frame_graph* g, uint32_t* out, uint32_t n)
{
    if (!g || (!out && n)) return RT_FAIL(g ? g->ctx : nullptr, "null argument");
    const uint32_t v[RT_FRAME_GRAPH_STATE_WORDS] = { g->bo.count ? g->bo_interval : 0u, g->bo.age, g->bo_sorted ? 1u : 0u, (g->direct[0] ? 1u : 0u) | (g->direct[1] ? 2u : 0u),
                                                     g->eager ? 1u : 0u, g->moved ? 1u : 0u, g->four_wide ? 1u : 0u, g->no_unsafe ? 1u : 0u };
    for (uint32_t k = 0; k < n; k++) out[k] = k < RT_FRAME_GRAPH_STATE_WORDS ? v[k] : 0u;
    return RT_SUCCESS;
}

int rt_frame_graph_launch(rt_frame_graph* g)
{
    if (!g) return RT_FAIL(nullptr, "frame graph is null");
    rt_ctx* ctx = g->ctx;
    rt_bvh* b = g->bvh;
    // the graph holds raw device addresses: refuse to replay it once one of them has been freed
    if (g->scratch_generation != ctx->scratch_generation)
        return RT_FAIL(ctx, "frame graph is stale: the context's deferred-ray buffers were reallocated by a larger trace; create a new graph");
    if (g->build_generation != b->build_generation || !b->built)
        return RT_FAIL(ctx, "frame graph is stale: the BVH was rebuilt since the graph was created; create a new graph");
    RT_ENTER(ctx);
    if (g->eager) {
        g->last_launch = 2;
        return frame_issue(b, g->scene, &g->cam, g->rank, g->nranks, g->layout, g->d_contacts, g->d_out, g->slots_per_rank);
    }
    // the per-camera records live in the BVH: if they are not this camera's (the camera moved, another position was traced on
    // this BVH in between, another strategy prepared other parts), replay the sequence that starts with the preparation
    const bool same = b->marked_valid && std::memcmp(b->marked_origin, g->origin, sizeof g->origin) == 0;
    const int which = (same && (b->marked_parts & g->parts) == g->parts) ? 0 : 1;
    const bool debug = ctx->graph_debug != 0;          // knob RTMI_GRAPH_DEBUG=1
    if (ctx->graph_debug == 2)
        std::fprintf(stderr, "rtmi graph %p: launch exec[%d] dirty %d parts %#x marked %#x same %d strategy %d%d/%u no_unsafe %d plane %#x origin %.9g %.9g %.9g\n", (void*)g, which,
                     (int)g->dirty[which][g->slot[which]], g->parts, b->marked_parts, (int)same, (int)g->packet, (int)g->quad, g->budget, (int)g->no_unsafe, g->plane_bits_known,
                     g->origin[0], g->origin[1], g->origin[2]);
    // the instance to launch: the one launched last if its arguments are current, else the other one (its last launch is the older
    // one); its arguments are re-set only behind that launch
    int k = g->slot[which];
    if (g->direct[which]) {
        // kernels only: issued as plain launches from the recorded argument copies (a launch copies its arguments when it is enqueued, so
        // refreshing the copies never touches a frame in flight: no second instance, no event, no hipGraphExecKernelNodeSetParams)
        k = 0;
        if (g->dirty[which][0] || g->dirty[which][1]) {
            const double t0 = debug ? now_us() : 0.0;
            FrameMode m = graph_mode(g, which, true);
            g->nodes[which].failed = false;
            g->nodes[which].args_only = true;
            const int rc = graph_issue(g, &m);
            g->nodes[which].args_only = false;
            if (rc != RT_SUCCESS) return RT_ERROR;
            g->dirty[which][0] = g->dirty[which][1] = false;
            if (debug) { g->host_update_us += now_us() - t0; g->host_updates++; }
        }
    } else
    if (g->dirty[which][k]) {
        k ^= 1;
        if (g->dirty[which][k]) {
            const double t0 = debug ? now_us() : 0.0;
            if (g->launched[which][k]) {
                if (g->has_event[which][k]) RT_HIP(ctx, hipEventSynchronize(g->done[which][k]));
                else RT_HIP(ctx, hipStreamSynchronize(ctx->stream));          // launched before the camera first moved: no event behind it
            }
            FrameMode m = graph_mode(g, which, true);
            g->nodes[which].failed = false;
            g->nodes[which].exec = g->exec[which][k];
            if (graph_issue(g, &m) != RT_SUCCESS) return RT_ERROR;
            g->dirty[which][k] = false;
            if (debug) { g->host_update_us += now_us() - t0; g->host_updates++; }
        }
    }
    const double t1 = debug ? now_us() : 0.0;
    if (g->direct[which]) {
        // kernels only: the recorded launches, in the order they were captured (= the order of the node kinds)
        for (int kind = 0; kind < RT_NODE_KINDS; kind++) {
            GraphNodes::Direct& d = g->nodes[which].direct[kind];
            if (!g->nodes[which].node[kind]) continue;
            void* params[1] = { (void*)d.args };
            RT_HIP(ctx, hipLaunchKernel(d.func, d.grid, d.block, params, d.lds, ctx->stream));
        }
    } else
    RT_HIP(ctx, hipGraphLaunch(g->exec[which][k], ctx->stream));
    g->has_event[which][k] = g->moved && !g->direct[which];
    if (g->has_event[which][k]) RT_HIP(ctx, hipEventRecord(g->done[which][k], ctx->stream));
    if (debug) { g->host_launch_us += now_us() - t1; g->host_launches++; }
    g->launched[which][k] = true;
    g->slot[which] = k;
    // the blocks' start order follows the costs the kernel counts: sorted behind the first replay, then every few frames
    // (a camera that stands still keeps its order: the interval doubles up to 1 024 frames and starts again when the camera moves)
    if (g->bo.count && g->four_wide && (!g->bo_sorted || ++g->bo.age >= g->bo_interval)) {
        block_order_refresh(ctx, g->bo);
        if (g->bo_sorted && g->bo_interval < 1024u) g->bo_interval <<= 1;
        g->bo_sorted = true;
    }
    if (which == 1) {
        std::memcpy(b->marked_origin, g->origin, sizeof g->origin);
        b->marked_parts = same ? (b->marked_parts | g->parts) : g->parts;
        b->marked_valid = true;
    }
    g->last_launch = which;
    ctx->trace_launched = true;                 // the next synchronising call looks at the kernels' error word
    return RT_SUCCESS;
}

void rt_frame_graph_destroy(rt_frame_graph* g)
{
    if (!g) return;
    if (g->host_launches && g->ctx->graph_debug)
        std::fprintf(stderr, "rtmi graph %p (n = %u): host time per argument update %.2f us (%llu), per launch %.2f us (%llu)\n", (void*)g, g->bvh->n,
                     g->host_updates ? g->host_update_us / g->host_updates : 0.0, (unsigned long long)g->host_updates,
                     g->host_launch_us / g->host_launches, (unsigned long long)g->host_launches);
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    for (int a = 0; a < 2; a++) {
        for (int k = 0; k < 2; k++) {
            if (g->exec[a][k]) (void)hipGraphExecDestroy(g->exec[a][k]);
            if (g->done[a][k]) (void)hipEventDestroy(g->done[a][k]);
        }
        if (g->graph[a]) (void)hipGraphDestroy(g->graph[a]);
    }
    block_order_free(g->bo);
    delete g;
}

}  // extern "C"
