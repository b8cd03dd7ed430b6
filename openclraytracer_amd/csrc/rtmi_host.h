// rtmi_host.h -- declarations shared by the host-side translation units of librtmi.so (rtmi_capi.hip, rtmi_bvh.hip,
// rtmi_trace_policy.hip, rtmi_frame_graph.hip, rtmi_rays.hip).  Not part of the ABI: hidden symbols.
#pragma once
#include "rtmi_internal.h"
#include <chrono>

#define RT_HIDDEN __attribute__((visibility("hidden")))
#define RT_BLOCK_ORDER_REFRESH 32u          // frames between two sorts of a frame shape's block order (one launch of one workgroup, k_block_order)

// Per-launch options of a primary frame beyond the strategy: a captured / updated frame graph, a fused colour buffer
struct FusedShade { void* rgba; uint32_t plane_stride; bool done; };
struct FrameMode {
    FusedShade* fused = nullptr;
    GraphNodes* record = nullptr;          // the launches are being captured: note the kernel nodes
    GraphNodes* update = nullptr;          // nothing is enqueued: set the arguments of captured nodes
    bool force_prepare = false;            // (re)compute the per-camera records whatever the BVH's marks say (the graph with the preparation in it)
    uint32_t plane_bits_known = 0;         // 0x80000000 | bits: the host knows the camera-on-a-plane bits
    uint32_t parts_drop = 0;               // RT_PREP_* this frame's kernels do not read although their strategy normally does
    // a captured frame keeps the strategy it was captured with (the BVH remembers one calibration, for the last resolution / tiling traced)
    bool fixed_strategy = false, packet = false, quad = false;
    uint32_t waves_per_block = 0;           // with fixed_strategy: the workgroup size the graph was captured with
    uint32_t budget = 0;
    BlockOrder* block_order = nullptr;     // a captured frame's own start-order buffers (null: the BVH's)
    bool eager_fused = false;              // rt_render_frame: an eager frame whose kernels shade what they store, whatever the strategy allows
};

extern "C" {
// rtmi_capi.hip
RT_HIDDEN int ensure_long_list(rt_ctx* ctx, uint64_t work_items);        // deferred-ray / split / key / task buffers for a launch of so many work items
RT_HIDDEN int check_trace_error(rt_ctx* ctx);                            // the trace kernels' sticky error word, once per launch batch
// rtmi_trace_policy.hip
RT_HIDDEN uint32_t strategy_parts(const rt_bvh* b, const rt_camera* cam, bool packet, bool quad);
RT_HIDDEN void block_order_free(BlockOrder& o);
RT_HIDDEN bool block_order_fit(rt_ctx* ctx, BlockOrder& o, uint32_t blocks);
RT_HIDDEN void block_order_refresh(rt_ctx* ctx, BlockOrder& o);
RT_HIDDEN TraceLaunch make_launch(rt_bvh* b, const rt_camera* cam = nullptr, uint32_t parts = 0, const FrameMode* mode = nullptr);
RT_HIDDEN int trace_primary_tuned(rt_bvh* b, const rt_camera* cam, uint32_t rank, uint32_t nranks, bool compact, rt_contact* out, const FrameMode* mode = nullptr);
RT_HIDDEN void current_strategy(const rt_bvh* b, bool& packet, bool& quad);
RT_HIDDEN int ensure_host_planes(rt_bvh* b);
RT_HIDDEN uint32_t host_plane_bits(const rt_bvh* b, const float origin[3]);
RT_HIDDEN const uint2* tile_table(rt_ctx* ctx, const rt_camera* cam, uint32_t rank, uint32_t nranks);
// rtmi_rays.hip
enum { RT_RAYKIND_FOREIGN = 0, RT_RAYKIND_SHADOW = 1, RT_RAYKIND_REFLECT = 2, RT_RAYKIND_PRIMARY = 3 };
RT_HIDDEN void note_ray_buffer(rt_ctx* ctx, const void* d_rays, int raykind);      // a generator of the library wrote this buffer
RT_HIDDEN void drop_rays_tuned(rt_bvh* b);                                          // forget the ray-buffer entry's calibrations (and their events)
}
// Deadlines of calibration launches on the device's own clock (trace_kernels.hip, cal_expired): a sample of that clock next to a host
// time stamp maps "now + limit" onto it.  The sample's kernel ran before the read-back returned, so mapped deadlines come a few
// tens of microseconds early: a slack is added.  A launch that was cut off says so in TraceCtrl::expired (candidate_state).
struct CalClock {
    unsigned long long clock0 = 0; double ticks_per_ms = 0.0; std::chrono::steady_clock::time_point host0;
    void sample(rt_ctx* ctx);                                     // behind whatever is enqueued on the context's stream; synchronises
    unsigned long long deadline(float limit_ms) const             // 0 = no deadline
    {
        if (!(ticks_per_ms > 0.0) || !(limit_ms > 0.f)) return 0ull;
        const double now_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host0).count();
        return clock0 + (unsigned long long)((now_ms + (double)limit_ms + 0.1) * ticks_per_ms);
    }
};
// after a calibration launch has finished: its error word (cleared again when set: a discarded candidate must not fail the frame) and
// whether a wave stopped at the deadline; false = a runtime call failed
RT_HIDDEN bool candidate_state(rt_ctx* ctx, uint32_t& error_bits, bool& expired);

static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
