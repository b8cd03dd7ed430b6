// rtmi_rays.hip -- C ABI, the ray-buffer entry point: generateContacts(CLBuffer& rays, CLBuffer& contacts, n), BVHManager.cpp:292-329
// (kernel generateContacts2, BVHKernels.cl:103-117; the entry the reference plans its secondary rays on, KNOWN_ISSUES_AND_TODO.txt:27-30).
//
// Which kernels walk a buffer is a speed matter only (every strategy gives the reference's contacts, trace_kernels.hip):
//   lane-per-ray (k_trace<0>) with a hand-over budget, then the wave-per-ray kernel for what is still walking -- right for rays that
//     have nothing to do with their neighbours;
//   four-wide packets (k_trace_packet4<0, ..>): 64 consecutive rays per wave, one record fetch through the scalar path per step for
//     all of them, sub-trees shared between waves under a step budget -- right where consecutive rays resemble each other (a camera's
//     rays in pixel order, shadow rays towards one light, reflections off a smooth surface).
// The first sufficiently large buffer of a SIZE CLASS (power of two) traced through a BVH calibrates the choice for that class (a handful
// of timed launches on that buffer, each under a deadline on the device clock; the results are the same bits, so the caller's contact
// buffer is simply written several times); kept across rebuilds of the same triangle count.  RTMI_RAYS=lane:<steps> | p4 | p4s:<steps> fixes it.
#include "rtmi_internal.h"
#include "rtmi_host.h"
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace {
struct RayStrategy { bool packet; uint32_t budget; };
#define RT_RAYS_CALIBRATE_FROM 32768u          // buffers below this are traced with what is known (default: lane-per-ray, 64 steps)

bool forced_strategy(const rt_ctx* ctx, RayStrategy& out)       // knob RTMI_RAYS (read in rt_ctx_create)
{
    out = { false, 64u };
    const char* e = ctx->rays_forced.c_str();
    bool set = false;
    if (std::strncmp(e, "lane:", 5) == 0) { set = true; out = { false, (uint32_t)std::strtoul(e + 5, nullptr, 10) }; }
    else if (std::strncmp(e, "p4s:", 4) == 0) { set = true; out = { true, (uint32_t)std::strtoul(e + 4, nullptr, 10) }; }
    else if (std::strcmp(e, "p4") == 0) { set = true; out = { true, 1u << 20 }; }
    if (set && out.budget == 0u) out.budget = 1u << 20;
    return set;
}

uint32_t size_class(uint32_t count) { uint32_t c = 0; while (count >>= 1) c++; return c; }       // floor(log2)
int ray_buffer_kind(const rt_ctx* ctx, const void* d_rays)
{
    for (const auto& t : ctx->ray_tags) if (t.addr == d_rays) return t.kind;
    return RT_RAYKIND_FOREIGN;
}
// the calibration key: size class + what the buffer holds (a generator's kind; a caller's own buffer: its address)
int find_tuned(const rt_bvh* b, uint32_t count, int kind, const void* d_rays)
{
    for (int i = 0; i < rt_bvh::RT_RAYS_TUNED; i++) {
        const rt_bvh::RaysTuned& e = b->rays_tuned[i];
        if (e.valid && e.size_class == size_class(count) && e.kind == (uint8_t)kind && (kind != RT_RAYKIND_FOREIGN || e.addr == d_rays)) return i;
    }
    return -1;
}
void drop_entry(rt_bvh::RaysTuned& e)
{
    if (e.e0) (void)hipEventDestroy(e.e0);
    if (e.e1) (void)hipEventDestroy(e.e1);
    e = rt_bvh::RaysTuned();
}
// The tuned launch before this one, if it has finished: did it take about what the calibration measured?  Two launches in a row at more than
// twice the calibrated time (+ 20 us) retire the entry: the buffer's rays are no longer what they were (never waits: hipEventQuery).
bool entry_went_stale(rt_bvh::RaysTuned& e)
{
    if (!e.pending || !e.e0 || !e.e1) return false;
    if (hipEventQuery(e.e1) != hipSuccess) { (void)hipGetLastError(); return false; }
    e.pending = false;
    float m = 0.f;
    if (hipEventElapsedTime(&m, e.e0, e.e1) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (m > 2.0f * e.ms + 0.02f) e.slow++; else e.slow = 0;
    return e.slow >= 2u;
}

void issue(rt_bvh* b, const RayStrategy& st, int tag, unsigned long long deadline, const rt_ray* rays, uint32_t count, rt_contact* out)
{
    rt_ctx* ctx = b->ctx;
    TraceLaunch t = make_launch(b);
    const bool four_wide = st.packet && b->d_quad && b->have_search_tree && b->d_trirec;
    t.packet = four_wide; t.packet4 = four_wide;
    t.defer_budget = st.budget;
    if (tag) { t.tag = tag; t.cal_deadline = deadline; }
    launch_trace_rays(ctx->stream, t, rays, count, out);
}

int calibrate(rt_bvh* b, const rt_ray* rays, uint32_t count, rt_contact* out, int kind)
{
    rt_ctx* ctx = b->ctx;
    static const RayStrategy cands[] = { { false, 64u }, { true, 64u }, { true, 1u << 20 }, { false, 16u }, { false, 256u },       // the coarse pass, in this order
                                         { false, 32u }, { false, 128u }, { false, 1024u }, { true, 32u }, { true, 256u }, { false, 8u } };
    enum { LANE_64 = 0, P4S_64, P4_WHOLE, LANE_16, LANE_256, LANE_32, LANE_128, LANE_1024, P4S_32, P4S_256, LANE_8, N_CAND };
    static_assert(sizeof cands / sizeof cands[0] == N_CAND, "candidate indices");
    const bool four_wide_ok = b->d_quad && b->have_search_tree && b->d_trirec;
    float ms[N_CAND]; int n[N_CAND];
    for (int i = 0; i < N_CAND; i++) { ms[i] = 0.f; n[i] = (cands[i].packet && (!four_wide_ok || (cands[i].budget < (1u << 20) && !ctx->d_task_list))) ? -1 : 0; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    RT_HIP(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return RT_FAIL(ctx, "hipEventCreate failed"); }
    CalClock clock;
    uint32_t launches = 0; float gpu_ms = 0.f; bool broken = false;
    auto best_of = [&](int skip) { int best = -1; for (int i = 0; i < N_CAND; i++) if (i != skip && n[i] > 0 && (best < 0 || ms[i] < ms[best])) best = i; return best; };
    auto sample = [&](int ci, bool timed) {
        if (broken || ci < 0 || ci >= N_CAND || n[ci] < 0) return;
        const int best = best_of(-1);
        const float limit = (timed && best >= 0) ? 1.6f * ms[best] + 0.02f : 0.f;
        float m = 0.f;
        (void)hipEventRecord(e0, ctx->stream);
        issue(b, cands[ci], 1, clock.deadline(limit), rays, count, out);
        (void)hipEventRecord(e1, ctx->stream);
        launches++;
        uint32_t err = 0; bool cut_off = false;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&m, e0, e1) != hipSuccess || !candidate_state(ctx, err, cut_off)) { broken = true; return; }
        gpu_ms += m;
        if (ctx->calibration_log)
            std::fprintf(stderr, "rtmi ray-buffer calibration: n=%u %u rays %s budget %u: %.4f ms%s%s\n", b->n, count, cands[ci].packet ? "packet4" : "lane-per-ray",
                         cands[ci].budget, m, err ? " (error word set, discarded)" : "", cut_off ? " (stopped at its deadline: no sample)" : "");
        if (!timed) return;
        if (err) { n[ci] = -1; return; }
        if (cut_off) { if (n[ci] == 0) n[ci] = -1; return; }
        ms[ci] = n[ci] == 0 ? m : (m < ms[ci] ? m : ms[ci]);
        n[ci]++;
    };
    sample(LANE_64, false);                     // untimed: clocks, code
    if (!broken) clock.sample(ctx);
    for (int ci : { (int)LANE_64, (int)P4S_64, (int)P4_WHOLE, (int)LANE_16, (int)LANE_256 }) sample(ci, true);
    switch (best_of(-1)) {                      // the neighbours of the best
    case LANE_16: sample(LANE_8, true); sample(LANE_32, true); break;
    case LANE_64: sample(LANE_32, true); sample(LANE_128, true); break;
    case LANE_256: sample(LANE_128, true); sample(LANE_1024, true); break;
    case P4S_64: sample(P4S_32, true); sample(P4S_256, true); break;
    default: break;
    }
    { const int a = best_of(-1), c = best_of(a); sample(a, true); sample(c, true); }        // the two fastest once more
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const int best = broken ? -1 : best_of(-1);
    if (best < 0) return RT_FAIL(ctx, "ray-buffer calibration: no strategy produced valid contacts");
    rt_bvh::RaysTuned& slot = b->rays_tuned[b->rays_tuned_next % rt_bvh::RT_RAYS_TUNED];
    b->rays_last = (int)(b->rays_tuned_next % rt_bvh::RT_RAYS_TUNED);
    b->rays_tuned_next++;
    drop_entry(slot);
    slot.size_class = size_class(count); slot.packet = cands[best].packet; slot.budget = cands[best].budget; slot.valid = true;
    slot.kind = (uint8_t)kind; slot.addr = (kind == RT_RAYKIND_FOREIGN ? (const void*)rays : nullptr); slot.ms = ms[best];
    if (hipEventCreate(&slot.e0) != hipSuccess || hipEventCreate(&slot.e1) != hipSuccess) {       // no events: the entry is never re-examined
        (void)hipGetLastError();
        if (slot.e0) (void)hipEventDestroy(slot.e0);
        slot.e0 = slot.e1 = nullptr;
    }
    b->rays_calibration_launches = launches; b->rays_calibration_gpu_ms = gpu_ms;
    return RT_SUCCESS;
}
}  // namespace

extern "C" {

void note_ray_buffer(rt_ctx* ctx, const void* d_rays, int raykind)
{
    if (!ctx || !d_rays) return;
    for (auto& t : ctx->ray_tags) if (t.addr == d_rays) { t.kind = (uint8_t)raykind; if (raykind == RT_RAYKIND_FOREIGN) t.addr = nullptr; return; }
    if (raykind == RT_RAYKIND_FOREIGN) return;
    auto& t = ctx->ray_tags[ctx->ray_tags_next++ % (sizeof ctx->ray_tags / sizeof ctx->ray_tags[0])];
    t.addr = d_rays; t.kind = (uint8_t)raykind;
}

void drop_rays_tuned(rt_bvh* b)
{
    for (auto& e : b->rays_tuned) drop_entry(e);
    b->rays_last = -1;
}

int rt_bvh_generate_contacts_rays(rt_bvh* b, const void* d_rays, uint32_t ray_count, void* d_contacts)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (ray_count == 0) return RT_SUCCESS;
    if (!d_rays || !d_contacts) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, ray_count)) return RT_ERROR;
    RayStrategy st = { false, 64u };
    rt_bvh::RaysTuned* timed = nullptr;
    if (ctx->defer_budget) st = { ctx->force_packet4, ctx->defer_budget };            // RTMI_DEFER_BUDGET (+ RTMI_PACKET=4): as for primary frames
    else if (!forced_strategy(ctx, st)) {
        if (ray_count >= RT_RAYS_CALIBRATE_FROM) {
            const int kind = ray_buffer_kind(ctx, d_rays);
            int i = find_tuned(b, ray_count, kind, d_rays);
            if (i >= 0 && entry_went_stale(b->rays_tuned[i])) { drop_entry(b->rays_tuned[i]); b->rays_recalibrations++; i = -1; }
            if (i < 0) { if (calibrate(b, (const rt_ray*)d_rays, ray_count, (rt_contact*)d_contacts, kind)) return RT_ERROR; i = b->rays_last; }
            b->rays_last = i;
            st = { b->rays_tuned[i].packet, b->rays_tuned[i].budget };
            if (b->rays_tuned[i].e0 && !b->rays_tuned[i].pending) timed = &b->rays_tuned[i];
        }
    }
    // (after a calibration: once more with the strategy chosen -- the last launch above was some other candidate's and may have been cut off)
    if (timed) (void)hipEventRecord(timed->e0, ctx->stream);
    issue(b, st, 0, 0ull, (const rt_ray*)d_rays, ray_count, (rt_contact*)d_contacts);
    if (timed) { (void)hipEventRecord(timed->e1, ctx->stream); timed->pending = true; }
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_bvh_rays_strategy(const rt_bvh* b, int* packet_kernel, uint32_t* budget, uint32_t* calibration_launches, float* calibration_gpu_ms)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    RayStrategy st = { false, 0u };
    if (b->ctx->defer_budget) st = { b->ctx->force_packet4, b->ctx->defer_budget };
    else if (!forced_strategy(b->ctx, st) && b->rays_last >= 0 && b->rays_tuned[b->rays_last].valid) st = { b->rays_tuned[b->rays_last].packet, b->rays_tuned[b->rays_last].budget };
    else if (!forced_strategy(b->ctx, st)) st = { false, 0u };
    if (packet_kernel) *packet_kernel = st.packet ? 2 : 0;
    if (budget) *budget = st.budget;
    if (calibration_launches) *calibration_launches = b->rays_calibration_launches;
    if (calibration_gpu_ms) *calibration_gpu_ms = b->rays_calibration_gpu_ms;
    return RT_SUCCESS;
}

int rt_bvh_rays_calibrations(const rt_bvh* b, uint32_t* entries, uint32_t* recalibrations)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    uint32_t n = 0;
    for (const auto& e : b->rays_tuned) n += e.valid ? 1u : 0u;
    if (entries) *entries = n;
    if (recalibrations) *recalibrations = b->rays_recalibrations;
    return RT_SUCCESS;
}

}  // extern "C"
