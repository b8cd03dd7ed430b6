// rtmi_trace_policy.hip -- which kernels trace a primary frame of a BVH, and with which budget: the first-frame calibration, the
// per-camera preparation a strategy needs, the start order of the blocks, the host's copy of the box planes; and the
// generateContacts(Camera&) entry points (BVHManager.cpp:234-284).  Every choice made here yields the same contacts.
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

void CalClock::sample(rt_ctx* ctx)
{
    ticks_per_ms = 0.0;
    if (ctx->no_calibration_deadline) return;          // knob RTMI_CALIBRATION_DEADLINE=0
    if (!ctx->d_clock && hipMalloc((void**)&ctx->d_clock, 8) != hipSuccess) { (void)hipGetLastError(); ctx->d_clock = nullptr; return; }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); return; }
    launch_device_clock(ctx->stream, ctx->d_clock);
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, ctx->d_clock, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    host0 = std::chrono::steady_clock::now();
    clock0 = v; ticks_per_ms = (double)khz;
}

bool candidate_state(rt_ctx* ctx, uint32_t& error_bits, bool& expired)
{
    // `expired` and the error word stand on neighbouring 128-byte lines of the control block: one copy
    const size_t lo = trace_ctrl_expired_offset(), hi = trace_ctrl_error_offset();
    uint32_t words[64] = { 0 };
    const size_t bytes = hi + 4 - lo;
    if (hi < lo || bytes > sizeof words) return false;
    if (hipMemcpyAsync(words, (const char*)ctx->d_trace_ctrl + lo, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    expired = words[0] != 0u;
    error_bits = words[(hi - lo) / 4];
    if (error_bits && (hipMemsetAsync((char*)ctx->d_trace_ctrl + hi, 0, 4, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) return false;
    return true;
}

extern "C" {

// Direction octants (bit k: 1 / direction negative on axis k) the rays of a camera can have: every component of the un-normalised
// direction (generateRay, CLStructs.h:95-106) is monotone in the pixel's x and in its y, so its extremes over the frame stand at the
// four corner pixels.  A superset is harmless (a copy of the four-wide records that nobody reads is written), a missing octant
// is not (the kernel raises error bit 7), hence the margin for the rounding of the fp32 evaluation.
static uint32_t camera_octants(const rt_camera* cam)
{
    const double w = (double)cam->resX, h = (double)cam->resY, fov = (double)cam->FOVDistance;
    const double cx[2] = { -(0.0 - w * 0.5), -((w - 1.0) - w * 0.5) }, cy[2] = { 0.0 - h * 0.5, (h - 1.0) - h * 0.5 };
    bool neg[3], pos[3];
    for (int k = 0; k < 3; k++) {
        const float* m = cam->viewTransform + 4 * k;
        neg[k] = pos[k] = false;
        for (int i = 0; i < 4; i++) {
            const double dx = cx[i & 1], dy = cy[i >> 1];
            const double v = dx * m[0] + dy * m[1] + fov * m[2] + m[3];
            const double mag = std::fabs(dx * m[0]) + std::fabs(dy * m[1]) + std::fabs(fov * m[2]) + std::fabs((double)m[3]);
            if (!(std::fabs(v) <= 1e300) || !(mag <= 1e300)) { neg[k] = pos[k] = true; break; }
            if (v - 1e-5 * mag <= 0.0) neg[k] = true;
            if (v + 1e-5 * mag >= 0.0) pos[k] = true;
        }
    }
    uint32_t mask = 0;
    for (uint32_t o = 0; o < 8u; o++) {
        bool can = true;
        for (int k = 0; k < 3; k++) can = can && ((o >> k) & 1u ? neg[k] : pos[k]);
        if (can) mask |= 1u << o;
    }
    return mask;
}

// what a first-phase kernel reads of the per-camera records (k_prepare_camera writes that and nothing else)
uint32_t strategy_parts(const rt_bvh* b, const rt_camera* cam, bool packet, bool quad)
{
    uint32_t parts = RT_PREP_MARKS;                    // ray_root's plane bits; the wave-per-ray / binary kernels' inside bits
    if (!b->d_quad || !b->have_search_tree) return parts;
    if (packet && quad) {
        parts |= RT_PREP_TRIANGLES;
        parts |= b->quad_cam_variants == 9u ? RT_PREP_OCTANTS(camera_octants(cam)) : RT_PREP_GENERAL;
    } else if (!packet) {
        parts |= RT_PREP_GENERAL;                      // the lane-per-ray kernel's four-wide step
    }
    return parts;
}
static uint32_t calibration_parts(const rt_bvh* b, const rt_camera* cam)
{
    return strategy_parts(b, cam, true, true) | strategy_parts(b, cam, false, false) | strategy_parts(b, cam, true, false);
}

// ---- start order of the blocks (BlockOrder, rtmi_internal.h) -------------------------------------------------------------------
void block_order_free(BlockOrder& o)
{
    if (o.d_cost) (void)hipFree(o.d_cost);
    if (o.d_order) (void)hipFree(o.d_order);
    o = BlockOrder();
}
// buffers for a frame of `blocks` blocks; false: none (allocation failed): the frame runs in index order
bool block_order_fit(rt_ctx* ctx, BlockOrder& o, uint32_t blocks)
{
    if (blocks == 0 || (blocks & 63u) || blocks > (4096u << 6)) return false;     // (what one launch of k_block_order sorts: frames up to 16.7 M pixels)
    if (blocks > o.cap) {
        (void)hipStreamSynchronize(ctx->stream);
        block_order_free(o);
        if (hipMalloc((void**)&o.d_cost, (size_t)blocks * 4) != hipSuccess || hipMalloc((void**)&o.d_order, (size_t)blocks * 4) != hipSuccess) { (void)hipGetLastError(); block_order_free(o); return false; }
        o.cap = blocks;
    }
    if (o.count != blocks) { o.count = blocks; o.valid = false; o.age = 0; }
    return true;
}
// the costs the frame just enqueued will have counted -> the order of the following frames (stream-ordered behind that frame)
void block_order_refresh(rt_ctx* ctx, BlockOrder& o)
{
    o.valid = launch_block_order(ctx->stream, o.d_cost, o.d_order, o.count, ctx->block_order_group_log, false);
    o.age = 0;
}

// Per-launch options of a primary frame beyond the strategy: a captured / updated frame graph, a fused colour buffer
TraceLaunch make_launch(rt_bvh* b, const rt_camera* cam, uint32_t parts, const FrameMode* mode)
{
    TraceLaunch t{};
    t.device_arith = b->device_arith;
    t.trav = b->d_trav; t.search_root = b->have_search_tree ? b->n - 1 : 0u; t.tree_select = b->d_tree_select; t.scene = &b->scene->d; t.pos = b->d_pos; t.trirec = b->d_trirec; t.tricam = b->d_tricam; t.n_leaves = b->n;
    t.ctrl = b->ctx->d_trace_ctrl; t.long_list = b->ctx->d_long_list; t.long_cap = b->ctx->long_cap;
    t.defer_budget = b->ctx->defer_budget;
    t.packet = false; t.packet4 = false; t.tag = b->ctx->profiling_tag; t.diagnostics = b->ctx->trace_diagnostics;
    t.quad = b->d_quad; t.quad_cam = b->d_quad_cam; t.quad_cam_variants = b->quad_cam_variants; t.no_unsafe_rays = b->capture_no_unsafe_rays;
    t.waves_per_block = (mode && mode->fixed_strategy) ? mode->waves_per_block
                        : b->ctx->trace_workgroup_waves ? b->ctx->trace_workgroup_waves : (b->ctx->frames_in_flight > 1u ? 1u : 0u);
    t.keys = b->ctx->d_keys; t.split_list = b->ctx->d_split_list; t.pairs = b->d_pairs;
    t.task_shards = b->ctx->d_task_list; t.task_cap = b->ctx->task_cap; t.task_helper_blocks = b->ctx->task_helper_blocks; t.long_batch = b->ctx->long_batch;
    t.packet_spill = b->d_packet_spill; t.packet_spill_entries = b->packet_spill_entries; t.packet_spill_slots = b->packet_spill_slots;
    t.ev[0] = t.ev[1] = t.ev[2] = nullptr;
    if (b->ctx->profiling && !(mode && mode->update)) {
        for (int k = 0; k < 3; k++) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            t.ev[k] = e;
            b->ctx->trace_events.push_back(e);
        }
    }
    t.compute_units = b->ctx->prop.multiProcessorCount;
    t.prepare_parts = 0;
    if (mode) { t.graph_record = mode->record; t.graph_update = mode->update; t.plane_bits_known = mode->plane_bits_known; }
    if (cam) {
        const float o[3] = { cam->viewTransform[3], cam->viewTransform[7], cam->viewTransform[11] };
        const bool same = b->marked_valid && std::memcmp(o, b->marked_origin, sizeof o) == 0;
        if (mode && (mode->update || mode->record)) {
            // a captured frame: its preparation node (if it has one) writes `parts` and its kernels read them.  The BVH's marks
            // are untouched here (nothing runs now): rt_frame_graph_launch sets them when it replays the graph with the preparation
            t.prepare_parts = mode->force_prepare ? parts : 0u;
            t.octant_mask = (parts >> 2) & 0xFFu;
        } else {
            const uint32_t have = same ? b->marked_parts : 0u;
            t.prepare_parts = parts & ~have;
            std::memcpy(b->marked_origin, o, sizeof o);
            b->marked_parts = have | parts;
            b->marked_valid = true;
            t.octant_mask = (b->marked_parts >> 2) & 0xFFu;
        }
    }
    if (!(mode && mode->update)) b->ctx->trace_launched = true;
    return t;
}

// Primary frames: which deferral budget is best depends on the scene (a few very long rays: small budget; many
// moderately long rays: no deferral), and every budget produces the same contacts.  The first frame of a
// (BVH, resolution, tiling) combination therefore runs once per candidate, timed with HIP events on the stream,
// and later frames use the fastest.  RTMI_DEFER_BUDGET fixes the budget and switches the calibration off.
// eager four-wide frames on one build before the host reads the leaf boxes' planes back (once per build: a synchronisation, three
// sorts on the device and their copies -- 11 ms at 70 000 leaves, which a scene that is rebuilt every frame or lives for a few
// frames never pays; it pays back 2-6 us per frame and the marks of a moved camera)
#ifndef RT_EAGER_PLANES_AFTER
#define RT_EAGER_PLANES_AFTER 32u
#endif

// the table work_to_pixel_block reads: (x0, y0) of tile `ordinal` of this rank for (resolution, rank, tiling); null when the context
// already holds RT_TILE_TABLES shapes or the allocation fails (the kernels then compute it per lane)
const uint2* tile_table(rt_ctx* ctx, const rt_camera* cam, uint32_t rank, uint32_t nranks)
{
    for (uint32_t i = 0; i < ctx->n_tile_tables; i++) {
        const rt_ctx::TileTable& e = ctx->tile_tables[i];
        if (e.resX == cam->resX && e.resY == cam->resY && e.rank == rank && e.nranks == nranks) return e.d;      // (the whole tuple: no hashed key)
    }
    if (ctx->n_tile_tables >= rt_ctx::RT_TILE_TABLES) return nullptr;
    const uint32_t tiles_x = rt_tiles_x(cam->resX), owned = rt_owned_tiles(tiles_x * rt_tiles_y(cam->resY), rank, nranks);
    if (owned == 0) return nullptr;
    std::vector<uint2> h(owned);
    for (uint32_t o = 0; o < owned; o++) { const uint32_t tile = rt_tile_of(rank, o, nranks); h[o] = make_uint2((tile % tiles_x) * RT_TILE, (tile / tiles_x) * RT_TILE); }
    uint2* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)owned * sizeof(uint2)) != hipSuccess || hipMemcpy(d, h.data(), (size_t)owned * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError(); if (d) (void)hipFree(d); return nullptr;
    }
    rt_ctx::TileTable& slot = ctx->tile_tables[ctx->n_tile_tables++];
    slot.resX = cam->resX; slot.resY = cam->resY; slot.rank = rank; slot.nranks = nranks; slot.d = d;
    return d;
}

int trace_primary_tuned(rt_bvh* b, const rt_camera* cam, uint32_t rank, uint32_t nranks, bool compact, rt_contact* out, const FrameMode* mode)
{
    rt_ctx* ctx = b->ctx;
    // start order of the blocks (four-wide kernel): the buffers of this frame shape
    BlockOrder* const bo = (mode && mode->block_order) ? mode->block_order : &b->block_order;
    const bool captured = mode && (mode->record || mode->update);
    const bool order_ok = !ctx->no_block_order && (ctx->frames_in_flight <= 1u || ctx->order_in_flight) && b->d_quad && b->have_search_tree &&
                          (captured ? bo->count != 0 : block_order_fit(ctx, *bo, (rt_tiles_owned_pixels(cam, rank, nranks) + 63u) / 64u));
    const uint64_t key = ((uint64_t)cam->resX << 40) ^ ((uint64_t)cam->resY << 16) ^ ((uint64_t)nranks << 8) ^ rank ^ (compact ? 1ull << 63 : 0) ^
                         (ctx->frames_in_flight > 1u ? 1ull << 62 : 0);      // (one strategy per policy, rt_ctx_set_frames_in_flight)
    if (ctx->defer_budget == 0 && (!b->tuned || b->tuned_key != key) && !(mode && mode->fixed_strategy)) {
        // another resolution / tiling than the last frame's: known from before?
        for (auto& e : b->tuned_cache)
            if (e.valid && e.key == key) {
                if (b->tuned) {                     // the current one changes places with it
                    rt_bvh::TunedEntry cur; cur.key = b->tuned_key; cur.budget = b->tuned_budget; cur.packet = b->tuned_packet; cur.quad = b->tuned_quad; cur.valid = true;
                    const rt_bvh::TunedEntry hit = e; e = cur;
                    b->tuned_key = hit.key; b->tuned_budget = hit.budget; b->tuned_packet = hit.packet; b->tuned_quad = hit.quad;
                } else {
                    b->tuned = true; b->tuned_key = e.key; b->tuned_budget = e.budget; b->tuned_packet = e.packet; b->tuned_quad = e.quad; e.valid = false;
                }
                break;
            }
    }
    if (ctx->defer_budget == 0 && (!b->tuned || b->tuned_key != key) && !(mode && mode->fixed_strategy)) {
        if (mode && (mode->record || mode->update)) return RT_FAIL(ctx, "a frame graph needs a calibrated strategy (trace one frame eagerly first)");
        // candidates: (packet kernel?, budget) -- budget = steps after which a block (packet kernel) or a ray (lane-per-ray
        // kernel) hands its remaining sub-trees to the wave-per-ray kernel
        static const struct { bool packet; uint32_t budget; bool quad; } candidates[] = {
            { true, 1u << 20, true }, { true, 32u, true }, { true, 64u, true }, { true, 128u, true }, { true, 256u, true }, { true, 1024u, true },   // four-wide: whole blocks / sub-trees shared after so many steps
            { true, 64u }, { true, 256u }, { true, 1024u }, { true, 4096u }, { true, 1u << 20 },
            { false, 8u }, { false, 12u }, { false, 16u }, { false, 24u }, { false, 32u }, { false, 48u }, { false, 64u }, { false, 96u }, { false, 128u }, { false, 192u }, { false, 256u }, { false, 512u }, { false, 1024u }, { false, 2048u }, { false, 1u << 20 } };
        enum { P4_NONE = 0, P4_32 = 1, P4_64 = 2, P4_128 = 3, P4_256 = 4, P4_1024 = 5, PK_64 = 6, PK_256 = 7, PK_4096 = 9, PK_NONE = 10, LANE_8 = 11, LANE_16 = 13, LANE_128 = 19, LANE_NONE = 25 };
        hipEvent_t e0 = nullptr, e1 = nullptr;
        RT_HIP(ctx, hipEventCreate(&e0));
        RT_HIP(ctx, hipEventCreate(&e1));
        constexpr int N_CAND = (int)(sizeof candidates / sizeof candidates[0]);
        static_assert(N_CAND == LANE_NONE + 1, "candidate indices");
        float cand_ms[N_CAND];            // fastest sample so far, handicap applied
        int cand_n[N_CAND];               // samples taken; -1: produced an invalid frame (or is not usable here)
        bool broken = false;              // a runtime call failed: give up
        bool last_failed = false;         // the most recent launch left an invalid frame behind
        uint32_t launches = 0;
        float gpu_ms = 0.f;               // device time of all calibration launches (events on the stream)
        const uint32_t cal_parts = calibration_parts(b, cam);
        CalClock cal_clock;                                            // deadlines of the candidates on the device's own clock
        float limit_ms = 0.f;                                          // 0 = the next launch has no deadline
        // one launch of one candidate; false = runtime failure
        // (cut_off: a wave stopped at the deadline, so the elapsed time is not the strategy's time)
        auto run_once = [&](int ci, float& ms, bool& valid, bool& cut_off) -> bool {
            const auto& cand = candidates[ci];
            TraceLaunch t = make_launch(b, cam, cal_parts);              // (the camera is prepared by the first of them)
            t.packet = cand.packet; t.packet4 = cand.quad;
            t.defer_budget = cand.budget;
            t.tag = 1;
            if (cand.packet && cand.quad && order_ok) { t.block_cost = bo->d_cost; t.block_order = bo->valid ? bo->d_order : nullptr; }
            t.cal_deadline = cal_clock.deadline(limit_ms);
            ms = 0.f;
            (void)hipEventRecord(e0, ctx->stream);
            t.tile_xy = tile_table(ctx, cam, rank, nranks);
            launch_trace_primary(ctx->stream, t, *cam, rank, nranks, compact, out);
            (void)hipEventRecord(e1, ctx->stream);
            launches++;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
            gpu_ms += ms;
            uint32_t cand_err = 0;
            if (!candidate_state(ctx, cand_err, cut_off)) return false;
            valid = cand_err == 0;                                       // e.g. a list too small for this strategy: not a candidate
            last_failed = !valid;
            if (ctx->calibration_log)
                std::fprintf(stderr, "rtmi calibration: n=%u %ux%u %s budget %u: %.4f ms%s%s\n", b->n, cam->resX, cam->resY,
                             cand.packet ? (cand.quad ? "packet4" : "packet") : "lane-per-ray", cand.budget, ms, valid ? "" : " (error word set, discarded)",
                             cut_off ? " (stopped at its deadline: no sample)" : "");
            return true;
        };
        // The candidates are timed with the device to themselves.  Sharing sub-trees between waves fills a launch's idle tail --
        // which other streams' launches fill as well when frames overlap (bench.py's five streams: ManFigure shared after 256
        // steps is 0.3 % faster alone and costs the five-frame step 8 %) -- and it costs queue traffic that grows as the budget
        // shrinks.  So a sharing candidate must win clearly: its time counts 3 % longer, plus 3 % per halving of the budget
        // below 1 024 steps.
        auto handicap = [&](int ci) -> float {
            const auto& cand = candidates[ci];
            if (!(cand.packet && cand.quad && cand.budget < (1u << 20))) return 1.0f;
            float h = 1.03f;
            for (uint32_t bgt = cand.budget; bgt < 1024u; bgt <<= 1) h += 0.03f;
            // several frames in flight (rt_ctx_set_frames_in_flight): the helper waves and queues of a sharing launch cost the OTHER frames
            // more than its own shorter tail gains (the bundled five: 0.284 ms per step with whole blocks, 0.316 with Tenryuu shared
            // after 64 steps, which alone is 0.15 against 0.19 ms) -- sharing must then win by far, as it does where whole blocks crawl
            if (ctx->frames_in_flight > 1u) h *= ctx->in_flight_share_handicap;      // (1.6; knob RTMI_IN_FLIGHT_SHARE_HANDICAP)
            return h;
        };
        auto usable = [&](int ci) {
            const auto& cand = candidates[ci];
            if (cand.packet && ctx->no_packet) return false;
            if (!cand.packet && ctx->force_packet) return false;
            if (cand.quad && (!b->d_quad || !b->have_search_tree)) return false;   // four-wide records are sound on the search tree only
            if (cand.quad && cand.budget < (1u << 20) && !ctx->d_task_list) return false;   // sub-tree sharing needs its queues
            if (cand.quad && cand.budget < ctx->share_min_budget) return false;             // knob RTMI_SHARE_MIN_BUDGET
            return true;
        };
        for (int ci = 0; ci < N_CAND; ci++) { cand_ms[ci] = 0.f; cand_n[ci] = usable(ci) ? 0 : -1; }
        // one more sample of a candidate: the fastest of its samples counts
        auto sample = [&](int ci) {
            if (broken || ci < 0 || ci >= N_CAND || cand_n[ci] < 0) return;
            // a candidate that needs more than 1.6 x the best time so far will not win: it stops there
            {
                float best = 0.f;
                for (int k = 0; k < N_CAND; k++) if (cand_n[k] > 0 && (best == 0.f || cand_ms[k] < best)) best = cand_ms[k];
                limit_ms = best > 0.f ? 1.6f * best + 0.02f : 0.f;
            }
            float m = 0.f; bool valid = false, cut_off = false;
            if (!run_once(ci, m, valid, cut_off)) { broken = true; return; }
            if (!valid) { cand_n[ci] = -1; return; }
            // a launch that was cut off lost against the limit it was given; its elapsed time says nothing else (in particular it is
            // not a fast sample when the deadline was mapped too early): a candidate with earlier samples keeps them
            if (cut_off) { if (cand_n[ci] == 0) cand_n[ci] = -1; return; }
            m *= handicap(ci);
            cand_ms[ci] = cand_n[ci] == 0 ? m : (m < cand_ms[ci] ? m : cand_ms[ci]);
            cand_n[ci]++;
        };
        auto best_of = [&](int skip) {
            int best = -1;
            for (int ci = 0; ci < N_CAND; ci++)
                if (ci != skip && cand_n[ci] > 0 && (best < 0 || cand_ms[ci] < cand_ms[best])) best = ci;
            return best;
        };
        // The first frame is bounded: about ten full-size launches instead of round 2's ~65 (25 candidates x 2 + warm-up + a
        // final round; 1.2 s of GPU time over the bench's six BVHs against 0.1-7.5 ms frames).  A coarse pass over few, widely
        // spaced candidates, one sample each; then the neighbours of the best; then the two fastest once more -- a 0.1 ms frame
        // cannot rank strategies 10 % apart on one sample, and the first launches run on idle clocks, hence an untimed
        // warm-up launch.  With four-wide records the binary packet kernel does not compete (round 2's full calibration never
        // picked it where the four-wide kernel was available); the lane-per-ray kernel does: ClumsyDragon's centre columns
        // walk ~80 records per ray against a mean of 3.6, and one wave per such ray (wave-per-ray kernel behind a lane-per-ray
        // budget of 16 steps) beats eight of them in one packet.  RTMI_CALIBRATE_ALL=1 restores the exhaustive pass over all 25.
        const bool calibrate_all = ctx->calibrate_all;
        // The first candidate runs without a deadline, twice (an untimed warm-up, then its sample), so it should never be a
        // hopeless one: sharing after 64 steps has no tail (since the search tree's boxes are tight a 1080p block takes 3-13 steps on
        // average and 50-900 at most: Tenryuu whole blocks 0.173 ms, shared after 256 / 128 / 64 / 32 steps 0.195 / 0.18 / 0.132 /
        // 0.14; ManFigure 0.09 / 0.108 from 64 up; ClumsyDragon 0.61 / 0.40 / 0.31 / 0.25 / 0.22 against 0.127 for lane-per-ray 16),
        // whole blocks and the lane-per-ray budgets can be several times behind.
        const int first_cand = cand_n[P4_64] == 0 ? P4_64 : cand_n[P4_NONE] == 0 ? P4_NONE : cand_n[LANE_128] == 0 ? LANE_128 : -1;
        {
            int warm = first_cand;
            for (int ci = 0; ci < N_CAND && warm < 0; ci++) if (cand_n[ci] == 0) warm = ci;
            if (warm >= 0) { float m; bool v = false, c = false; if (!run_once(warm, m, v, c)) broken = true; }
            // the warm-up ran to its end: its block costs give the candidates the start order the frames will have
            if (!broken && warm >= 0 && candidates[warm].packet && candidates[warm].quad && order_ok) block_order_refresh(ctx, *bo);
            if (!broken) cal_clock.sample(ctx);
        }
        auto refine_lane = [&](int first) {
            // the neighbours of the best lane-per-ray budget, two candidates away on either side (the list is ~sqrt(2) apart)
            sample(first - 2 >= LANE_8 ? first - 2 : LANE_8);
            sample(first + 2 < LANE_NONE ? first + 2 : LANE_NONE - 1);
            // ... then the direct neighbours of whichever is best now (ClumsyDragon: 8 / 16 / 32 steps make 16 the best, 12 is 5 % better)
            const int mid = best_of(-1);
            if (mid >= LANE_8 && mid < LANE_NONE) {
                if (mid - 1 >= LANE_8 && cand_n[mid - 1] == 0) sample(mid - 1);
                if (mid + 1 < LANE_NONE && cand_n[mid + 1] == 0) sample(mid + 1);
            }
        };
        if (calibrate_all) {
            for (int ci = 0; ci < N_CAND; ci++) { sample(ci); sample(ci); }
            for (int round = 0, a = best_of(-1); round < 2 && a >= 0; round++) { const int c = best_of(a); sample(a); sample(c); a = best_of(-1); }
        } else if (cand_n[P4_NONE] == 0) {
            // coarse: one of each family -- sub-trees shared, whole blocks, lane-per-ray with an early hand-over
            for (int ci : { first_cand, (int)P4_NONE, (int)LANE_16 })
                if (ci >= 0 && cand_n[ci] == 0) sample(ci);
            // Whole blocks are the strategy that costs the other frames in flight nothing, and one slow sample of it -- the first launches of a
            // process run on idle clocks and an unsorted block order -- hands the scene to a sharing budget for the life of the BVH (seen once in
            // a profile round: Tenryuu shared after 32 steps under the throughput policy, the five-scene step 0.302 instead of 0.245 ms): before
            // whole blocks lose the coarse pass they get a second sample.
            if (best_of(-1) != P4_NONE && cand_n[P4_NONE] == 1) sample(P4_NONE);
            const int first = best_of(-1);
            if (first == P4_64) { sample(P4_32); sample(P4_128); if (best_of(-1) == P4_128) sample(P4_256); }
            else if (first >= LANE_8 && first < LANE_NONE) refine_lane(first);
            // (whole blocks won: a sharing budget lost with its handicap, the others would too)
            const int a = best_of(-1), c = best_of(a);
            sample(a); sample(c);
        } else {
            // no four-wide records (LITERAL_ZERO mode, RTMI_SEARCH_TREE=0 / RTMI_PACKET4=0): lane-per-ray budgets and the binary packet kernel
            for (int ci : { first_cand, (int)LANE_16, (int)LANE_128, (int)LANE_NONE, (int)PK_256, (int)PK_NONE })
                if (ci >= 0 && cand_n[ci] == 0) sample(ci);
            const int first = best_of(-1);
            if (first >= LANE_8 && first < LANE_NONE) refine_lane(first);
            else if (first >= PK_64 && first < LANE_8) {
                sample(first == PK_NONE ? (int)PK_4096 : first + 1);
                if (first > PK_64) sample(first - 1);
            }
            const int a = best_of(-1), c = best_of(a);
            sample(a); sample(c);
        }
        int best_ci = broken ? -1 : best_of(-1);
        // Several frames in flight: a lane-per-ray frame is two dependent kernels, and the second one -- a wave per handed-over ray -- is as long as
        // its heaviest ray whatever else the device does, while the first one's extra steps overlap with the other frames.  Between two budgets
        // that are within 8 % of each other alone, the larger one (fewer rays handed over) is the better neighbour: ClumsyDragon 12 -> 16 steps,
        // 0.081 -> 0.0845 ms alone, the five-scene step 0.231 -> 0.226 ms (tools/tail_probe.py).
        if (ctx->frames_in_flight > 1u && best_ci >= LANE_8 && best_ci + 1 < LANE_NONE && cand_n[best_ci + 1] > 0 &&
            cand_ms[best_ci + 1] <= 1.08f * cand_ms[best_ci]) best_ci++;
        const bool have = best_ci >= 0;
        const uint32_t best = have ? candidates[best_ci].budget : 1024u;
        const bool best_packet = have && candidates[best_ci].packet, best_quad = have && candidates[best_ci].quad;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!have) return RT_FAIL(ctx, "trace calibration: no strategy produced a valid frame");
        if (b->tuned) {                            // the calibration it replaces is kept for later
            rt_bvh::TunedEntry& slot = b->tuned_cache[b->tuned_cache_next++ % 4u];
            slot.key = b->tuned_key; slot.budget = b->tuned_budget; slot.packet = b->tuned_packet; slot.quad = b->tuned_quad; slot.valid = true;
        }
        b->tuned = true; b->tuned_key = key; b->tuned_budget = best; b->tuned_packet = best_packet; b->tuned_quad = best_quad;
        b->calibration_launches = launches; b->calibration_gpu_ms = gpu_ms;
        (void)last_failed;
        RT_HIP(ctx, hipGetLastError());
        // fall through: trace once more with the strategy just chosen (the last launch above was some other candidate's)
    }
    bool packet = false, quad = false; uint32_t budget = ctx->defer_budget;
    if (ctx->defer_budget == 0 && b->tuned) { budget = b->tuned_budget; packet = b->tuned_packet; quad = b->tuned_quad; }
    if (ctx->defer_budget != 0 && ctx->force_packet) { packet = true; quad = ctx->force_packet4 && b->d_quad; }
    if (mode && mode->fixed_strategy) { packet = mode->packet; quad = mode->quad; budget = mode->budget; }
    uint32_t parts = strategy_parts(b, cam, packet, quad) & ~(mode ? mode->parts_drop : 0u);
    // Eager frames of the four-wide kernel end with a launch of the wave-per-ray kernel that finds nothing to do unless the camera
    // sits on a box plane (~6.5 us of a 0.05-0.1 ms frame; a captured frame leaves it out, rt_frame_graph_create).  Once a build has
    // served a few frames the host knows the planes (ensure_host_planes: what rt_frame_graph_set_camera uses) and looks the camera up
    // itself: off every plane that matters, the launch and the per-camera marks are left out here as well.  Error bit 6 guards it.
    uint32_t eager_bits = 0; bool eager_no_unsafe = false;
    if (!captured && packet && quad && b->have_search_tree && b->d_quad && b->d_tricam) {
        if (b->eager_frames < RT_EAGER_PLANES_AFTER) b->eager_frames++;
        else if (ensure_host_planes(b) == RT_SUCCESS) {
            const float origin[3] = { cam->viewTransform[3], cam->viewTransform[7], cam->viewTransform[11] };
            const uint32_t bits = host_plane_bits(b, origin);
            eager_bits = 0x80000000u | bits;
            if ((bits & (b->quad_cam_variants == 9u ? 6u : 7u)) == 0u) { eager_no_unsafe = true; parts &= ~RT_PREP_MARKS; }
        }
    }
    TraceLaunch t = make_launch(b, cam, parts, mode);
    if (eager_bits) { t.plane_bits_known = eager_bits; t.no_unsafe_rays = eager_no_unsafe; }
    t.defer_budget = budget; t.packet = packet; t.packet4 = quad;
    if (mode && mode->fused) { t.fused_rgba = mode->fused->rgba; t.fused_plane_stride = mode->fused->plane_stride; t.fused_done = &mode->fused->done; }
    const bool ordered = packet && quad && order_ok;
    if (ordered) { t.block_cost = bo->d_cost; t.block_order = (bo->valid || captured) ? bo->d_order : nullptr; }
    t.tile_xy = tile_table(ctx, cam, rank, nranks);
    launch_trace_primary(ctx->stream, t, *cam, rank, nranks, compact, out);
    if (mode && mode->update) return mode->update->failed ? RT_FAIL(ctx, "frame graph: a kernel node could not be updated") : RT_SUCCESS;
    // every few frames the order follows the costs the frame just enqueued will have counted (a captured frame: rt_frame_graph_launch)
    if (ordered && !captured && (!bo->valid || ++bo->age >= RT_BLOCK_ORDER_REFRESH)) block_order_refresh(ctx, *bo);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_bvh_generate_contacts_primary(rt_bvh* b, const rt_camera* cam, void* d_contacts)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (!d_contacts) return RT_FAIL(ctx, "contacts buffer is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, (uint64_t)rt_tiles_owned_pixels(cam, 0, 1))) return RT_ERROR;
    return trace_primary_tuned(b, cam, 0, 1, false, (rt_contact*)d_contacts);
}

int rt_bvh_generate_contacts_primary_tiled(rt_bvh* b, const rt_camera* cam, uint32_t rank, uint32_t nranks,
                                           void* d_contacts_compact)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (!d_contacts_compact) return RT_FAIL(ctx, "contacts buffer is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, (uint64_t)rt_tiles_owned_pixels(cam, rank, nranks))) return RT_ERROR;
    return trace_primary_tuned(b, cam, rank, nranks, true, (rt_contact*)d_contacts_compact);
}

int rt_bvh_trace_strategy(const rt_bvh* b, int* packet_kernel, uint32_t* budget)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    const rt_ctx* ctx = b->ctx;
    if (budget) *budget = ctx->defer_budget ? ctx->defer_budget : (b->tuned ? b->tuned_budget : 0u);
    if (packet_kernel) *packet_kernel = ctx->defer_budget ? (ctx->force_packet ? (ctx->force_packet4 && b->d_quad ? 2 : 1) : 0)
                                                          : (b->tuned && b->tuned_packet ? (b->tuned_quad ? 2 : 1) : 0);
    return RT_SUCCESS;
}

int rt_bvh_block_order(rt_bvh* b, uint32_t* h_order, uint32_t* h_cost, uint32_t capacity, uint32_t* blocks)
{
    if (!b || !blocks) return RT_FAIL(b ? b->ctx : nullptr, "null argument");
    rt_ctx* ctx = b->ctx;
    const BlockOrder& o = b->block_order;
    *blocks = (o.valid && !ctx->no_block_order && (ctx->frames_in_flight <= 1u || ctx->order_in_flight)) ? o.count : 0u;
    if (*blocks == 0) return RT_SUCCESS;
    if (*blocks > capacity) return RT_FAIL(ctx, "rt_bvh_block_order: capacity too small");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_order) RT_HIP(ctx, hipMemcpy(h_order, o.d_order, (size_t)o.count * 4, hipMemcpyDeviceToHost));
    if (h_cost) RT_HIP(ctx, hipMemcpy(h_cost, o.d_cost, (size_t)o.count * 4, hipMemcpyDeviceToHost));
    return RT_SUCCESS;
}

uint32_t rt_bvh_calibration_launches(const rt_bvh* b) { return b ? b->calibration_launches : 0u; }
uint32_t rt_bvh_host_plane_reads(const rt_bvh* b) { return b ? b->h_planes_reads : 0u; }
float rt_bvh_calibration_gpu_ms(const rt_bvh* b) { return b ? b->calibration_gpu_ms : 0.f; }

int rt_bvh_search_tree(const rt_bvh* b, uint32_t* depth, int* used_by_last_camera)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (depth) *depth = b->have_search_tree ? b->search_depth : 0u;
    if (used_by_last_camera) {
        *used_by_last_camera = 0;
        if (b->have_search_tree && b->built && b->marked_valid) {
            uint32_t sel = 7;
            RT_ENTER(ctx);
            RT_HIP(ctx, hipMemcpyAsync(&sel, b->d_tree_select, 4, hipMemcpyDeviceToHost, ctx->stream));
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            *used_by_last_camera = sel == 0u ? 1 : 2;      // 2: only rays without a zero direction component on the camera's plane axes
        }
    }
    return RT_SUCCESS;
}

void current_strategy(const rt_bvh* b, bool& packet, bool& quad)
{
    const rt_ctx* ctx = b->ctx;
    packet = false; quad = false;
    if (ctx->defer_budget == 0 && b->tuned) { packet = b->tuned_packet; quad = b->tuned_quad; }
    if (ctx->defer_budget != 0 && ctx->force_packet) { packet = true; quad = ctx->force_packet4 && b->d_quad; }
    if (quad && (!b->d_quad || !b->d_tricam || !b->have_search_tree)) quad = false;        // (launch_tagged's own condition)
}

// "Does this camera coordinate equal a plane of a box" answered on the host: every box of either tree is a union of leaf boxes,
// clamped against the FLT_MIN the reference starts its maxima from (BVH.h:310), so the planes of all boxes are among the planes of
// the leaf boxes and FLT_MIN -- a superset of what k_prepare_camera finds on the device, which is the safe side.
int ensure_host_planes(rt_bvh* b)
{
    rt_ctx* ctx = b->ctx;
    if (b->h_planes_generation == b->build_generation && !b->h_planes[0].empty()) return RT_SUCCESS;
    // sorted on the device (launch_plane_keys), copied back per axis; FLT_MIN joins them here (BVH.h:310)
    const bool dbg_t = ctx->graph_debug != 0;
    const double t_begin = dbg_t ? now_us() : 0.0;
    uint32_t p2 = 1; while (p2 < 2u * b->n) p2 <<= 1;
    uint2* d_tmp = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&d_tmp, (size_t)p2 * sizeof(uint2)));
    std::vector<uint2> h(p2);
    for (uint32_t k = 0; k < 3; k++) {
        launch_plane_keys(ctx->stream, b->d_nodes, b->n, k, d_tmp, p2);
        hipError_t e = hipMemcpyAsync(h.data(), d_tmp, (size_t)p2 * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { (void)hipFree(d_tmp); return RT_FAIL(ctx, std::string("leaf planes: ") + hipGetErrorString(e)); }
        std::vector<float>& v = b->h_planes[k];
        v.clear(); v.reserve(2 * (size_t)b->n + 1);
        bool fmin_in = false;
        for (uint32_t i = 0; i < p2 && h[i].x != 0xFFFFFFFFu; i++) {
            const uint32_t key = h[i].x, bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
            float f; std::memcpy(&f, &bits, 4);
            if (!fmin_in && f > FLT_MIN) { if (v.empty() || v.back() != FLT_MIN) v.push_back(FLT_MIN); fmin_in = true; }
            if (v.empty() || v.back() != f) v.push_back(f);               // ascending: duplicates are neighbours (-0 == +0)
        }
        if (!fmin_in && (v.empty() || v.back() != FLT_MIN)) v.push_back(FLT_MIN);
    }
    (void)hipFree(d_tmp);
    if (dbg_t) std::fprintf(stderr, "rtmi: leaf planes of %u leaves on the host in %.0f us\n", b->n, now_us() - t_begin);
    b->h_planes_generation = b->build_generation;
    b->h_planes_reads++;
    return RT_SUCCESS;
}
uint32_t host_plane_bits(const rt_bvh* b, const float origin[3])
{
    uint32_t bits = 0;
    for (int k = 0; k < 3; k++)
        if (origin[k] == origin[k] && std::binary_search(b->h_planes[k].begin(), b->h_planes[k].end(), origin[k])) bits |= 1u << k;
    return bits;
}

}  // extern "C"
