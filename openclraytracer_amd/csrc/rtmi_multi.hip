// rtmi_multi.hip -- the multi-GPU frame behind the C ABI (include/rtmi.h, rt_comm_* / rt_multi_frame_*).
//
// New functionality: the reference drives exactly one OpenCL device (CLProgram::compile builds for one device,
// OpenCLUtils/CLExecutionContext.cpp:193-196).  Rays are independent and the scene + BVH are replicated, so the
// frame is cut into 64x64-pixel tiles dealt over the ranks (rtmi_internal.h, rt_tile_owner); every rank traces and shades its
// tiles into three packed float planes (rt_shade_blinn_phong_tiled_packed's 12-byte wire format) and ONE gather brings the
// planes to rank 0, which scatters them back to pixel order.  The gather is RCCL point-to-point, grouped: ncclRecv from every
// peer on the root, one ncclSend on each peer -- N-1 concurrent transfers, one per xGMI link of the root (a ring would be bound
// by a single link).  Nothing of this path runs in Python: one call renders a frame.
//
// One communicator rank per context.  Two ways to form a group:
//   rt_comm_create        one process per GPU (the launcher hands every rank the 128-byte id of rt_comm_unique_id) -- ncclCommInitRank
//   rt_comm_create_group  one process, N contexts -- ncclCommInitAll over their devices.  When contexts share a device (RCCL refuses
//                         two ranks on one device) the group exchanges by device-to-device copies instead: the transport of the
//                         one-GPU tests, everything else -- tiling, double buffering, stream order, untile -- being the same code.
// RCCL is loaded at run time (dlopen "librccl.so.1"): a process that already holds a copy (torch ships one) keeps using that one.
#include "rtmi_internal.h"

#include <dlfcn.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

// ---- the handful of RCCL entry points used, with the types of rccl.h (rccl/rccl.h:40-43,187-260,700-933) ------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                 // ncclSuccess = 0
enum { rtNcclFloat32 = 7 };               // ncclFloat32, rccl.h ncclDataType_t
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static const char* load_rccl()
{
    if (g_rccl.lib) return nullptr;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return "RCCL (librccl.so.1) cannot be loaded";
    RcclApi a;
    a.lib = h;
#define RT_SYM(field, name) *(void**)(&a.field) = dlsym(h, name); if (!a.field) return "RCCL lacks " name;
    RT_SYM(GetUniqueId, "ncclGetUniqueId") RT_SYM(CommInitRank, "ncclCommInitRank") RT_SYM(CommInitAll, "ncclCommInitAll")
    RT_SYM(CommDestroy, "ncclCommDestroy") RT_SYM(CommCount, "ncclCommCount") RT_SYM(Send, "ncclSend") RT_SYM(Recv, "ncclRecv") RT_SYM(GroupStart, "ncclGroupStart")
    RT_SYM(GroupEnd, "ncclGroupEnd") RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
    g_rccl = a;
    return nullptr;
}

#define RT_NCCL(ctx, expr)                                                                                       \
    do {                                                                                                         \
        ncclResult_t r__ = (expr);                                                                               \
        if (r__ != 0) return RT_FAIL((ctx), std::string(#expr) + ": " + g_rccl.GetErrorString(r__));            \
    } while (0)

struct rt_comm {
    rt_ctx* ctx = nullptr;
    int rank = 0, nranks = 1;
    ncclComm_t nccl = nullptr;             // null: the group exchanges by device copies (contexts share a device)
    hipStream_t stream = nullptr;          // the gather of a frame runs here, behind an event of the context's stream
    std::vector<rt_comm*> group;           // single-process groups: every rank's communicator, by rank (empty: one process per GPU)
    bool by_copies = false;
    bool owns_nccl = true;                 // false: rt_comm_share'd from another communicator of this process
    bool broken = false;                   // a gather failed half way: the ranks are out of step, every later render fails at once
};

struct rt_multi_frame {
    rt_comm* comm = nullptr;
    rt_bvh* bvh = nullptr;
    const rt_scene* scene = nullptr;
    rt_camera cam{};
    uint32_t tiling = 0;                   // RT_TILING word
    uint32_t own_slots = 0;                // slots of this rank's compact buffers
    uint32_t root_slots = 0, peer_slots = 0;   // plane strides: rank 0's planes / every peer's planes (the most any peer owns)
    void* d_contacts = nullptr;            // own_slots contacts
    float* d_send[2] = { nullptr, nullptr };   // three planes each, double buffered
    float* d_peers = nullptr;              // root: (nranks - 1) x 3 x peer_slots floats
    float4* d_frame = nullptr;             // root: the assembled frame
    rt_frame_graph* graph[2] = { nullptr, nullptr };
    hipEvent_t traced[2] = { nullptr, nullptr }, gathered[2] = { nullptr, nullptr };
    bool gathered_valid[2] = { false, false };
    uint64_t frames = 0;
    // which rank assembles this frame, and this rank's place counted from it: the tiles are dealt by VIRTUAL rank (the root is virtual
    // rank 0 and owns the extra share), so several frames in flight can gather into different GPUs (bench.py: scene i -> rank i mod N)
    int root = 0, vrank = 0;
    // optional timing (rt_multi_frame_set_timing): events around this rank's trace + shade and around the gather (+ untile on the root)
    bool timing = false;
    hipEvent_t tev[2][4] = { { nullptr, nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr, nullptr } };
    bool tev_pending[2] = { false, false };
    double trace_ms_sum = 0.0, gather_ms_sum = 0.0;
    uint64_t timed_frames = 0;
};

static void harvest_timing(rt_multi_frame* f, int b)
{
    if (!f->tev_pending[b]) return;
    float a = 0.f, g = 0.f;
    if (hipEventSynchronize(f->tev[b][3]) == hipSuccess && hipEventElapsedTime(&a, f->tev[b][0], f->tev[b][1]) == hipSuccess &&
        hipEventElapsedTime(&g, f->tev[b][2], f->tev[b][3]) == hipSuccess) {
        f->trace_ms_sum += a; f->gather_ms_sum += g; f->timed_frames++;
    }
    (void)hipGetLastError();
    f->tev_pending[b] = false;
}

extern "C" {

int rt_comm_unique_id(void* id128)
{
    if (!id128) return RT_FAIL(nullptr, "id buffer is null");
    if (const char* e = load_rccl()) return RT_FAIL(nullptr, e);
    ncclUniqueId id;
    RT_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    std::memcpy(id128, id.internal, sizeof id.internal);
    return RT_SUCCESS;
}

static int comm_alloc(rt_ctx* ctx, int rank, int nranks, rt_comm** out)
{
    rt_comm* c = new (std::nothrow) rt_comm();
    if (!c) return RT_FAIL(ctx, "out of host memory");
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return RT_FAIL(ctx, std::string("communicator stream: ") + hipGetErrorString(e)); }
    *out = c;
    return RT_SUCCESS;
}

int rt_comm_create(rt_ctx* ctx, const void* id128, int rank, int nranks, rt_comm** out)
{
    if (!ctx || !out) return RT_FAIL(ctx, "null argument");
    *out = nullptr;
    if (nranks < 1 || nranks > 0xFFFF || rank < 0 || rank >= nranks) return RT_FAIL(ctx, "rank/nranks invalid");
    if (nranks > 1 && !id128) return RT_FAIL(ctx, "the communicator id is null");
    rt_comm* c = nullptr;
    if (comm_alloc(ctx, rank, nranks, &c)) return RT_ERROR;
    if (nranks > 1) {
        if (const char* e = load_rccl()) { rt_comm_destroy(c); return RT_FAIL(ctx, e); }
        ncclUniqueId id;
        std::memcpy(id.internal, id128, sizeof id.internal);
        ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);      // collective over all ranks' processes
        if (r != 0) { std::string m = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); rt_comm_destroy(c); return RT_FAIL(ctx, m); }
    }
    *out = c;
    return RT_SUCCESS;
}

int rt_comm_create_group(rt_ctx** ctxs, int nranks, rt_comm** out)
{
    if (!ctxs || !out || nranks < 1 || nranks > 0xFFFF) return RT_FAIL(nullptr, "invalid argument");
    for (int r = 0; r < nranks; r++) { out[r] = nullptr; if (!ctxs[r]) return RT_FAIL(nullptr, "a context is null"); }
    bool shared = false;
    for (int a = 0; a < nranks; a++) for (int b = a + 1; b < nranks; b++) shared = shared || ctxs[a]->device == ctxs[b]->device;
    std::vector<rt_comm*> all((size_t)nranks, nullptr);
    for (int r = 0; r < nranks; r++)
        if (comm_alloc(ctxs[r], r, nranks, &all[r])) { for (rt_comm* c : all) rt_comm_destroy(c); return RT_ERROR; }
    if (nranks > 1 && !shared) {
        if (const char* e = load_rccl()) { for (rt_comm* c : all) rt_comm_destroy(c); return RT_FAIL(ctxs[0], e); }
        std::vector<int> devs; std::vector<ncclComm_t> comms((size_t)nranks, nullptr);
        for (int r = 0; r < nranks; r++) devs.push_back(ctxs[r]->device);
        ncclResult_t rc = g_rccl.CommInitAll(comms.data(), nranks, devs.data());
        if (rc != 0) { std::string m = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc); for (rt_comm* c : all) rt_comm_destroy(c); return RT_FAIL(ctxs[0], m); }
        for (int r = 0; r < nranks; r++) all[r]->nccl = comms[r];
    }
    for (int r = 0; r < nranks; r++) { all[r]->group = all; all[r]->by_copies = nranks > 1 && shared; out[r] = all[r]; }
    return RT_SUCCESS;
}

void rt_comm_destroy(rt_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->nccl && c->owns_nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Several frames in flight per rank (a renderer with several viewports; bench.py's five scenes): another context of the same
// process and device joins the owner's RCCL communicator with a stream of its own.  RCCL serialises operations on one
// communicator, so every rank must issue its frames in the same order.  The owner must outlive the sharer.
int rt_comm_share(const rt_comm* owner, rt_ctx* ctx, rt_comm** out)
{
    if (!owner || !ctx || !out) return RT_FAIL(ctx, "null argument");
    *out = nullptr;
    if (!owner->group.empty()) return RT_FAIL(ctx, "communicators of a single-process group are not shared: form another group");
    if (owner->ctx->device != ctx->device) return RT_FAIL(ctx, "a shared communicator must stay on its device");
    rt_comm* c = nullptr;
    if (comm_alloc(ctx, owner->rank, owner->nranks, &c)) return RT_ERROR;
    c->nccl = owner->nccl; c->owns_nccl = false;
    *out = c;
    return RT_SUCCESS;
}

int rt_comm_rank(const rt_comm* c) { return c ? c->rank : -1; }
int rt_comm_size(const rt_comm* c) { return c ? c->nranks : 0; }
int rt_comm_uses_rccl(const rt_comm* c) { return c && c->nccl ? 1 : 0; }
int rt_comm_rccl_ranks(const rt_comm* c)
{
    int n = 0;
    if (!c || !c->nccl || !g_rccl.CommCount || g_rccl.CommCount(c->nccl, &n) != 0) return 0;
    return n;                                      // what RCCL itself says the communicator spans
}

// ---------------------------------------------------------------------------------------------------------------------------
void rt_multi_frame_destroy(rt_multi_frame* f)
{
    if (!f) return;
    rt_ctx* ctx = f->comm->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(f->comm->stream);
    for (int b = 0; b < 2; b++) {
        if (f->graph[b]) rt_frame_graph_destroy(f->graph[b]);
        if (f->traced[b]) (void)hipEventDestroy(f->traced[b]);
        if (f->gathered[b]) (void)hipEventDestroy(f->gathered[b]);
        if (f->d_send[b]) (void)hipFree(f->d_send[b]);
        for (int k = 0; k < 4; k++) if (f->tev[b][k]) (void)hipEventDestroy(f->tev[b][k]);
    }
    if (f->d_contacts) (void)hipFree(f->d_contacts);
    if (f->d_peers) (void)hipFree(f->d_peers);
    if (f->d_frame) (void)hipFree(f->d_frame);
    delete f;
}

int rt_multi_frame_create(rt_comm* comm, rt_bvh* bvh, const rt_scene* scene, const rt_camera* cam, uint32_t root_extra, rt_multi_frame** out)
{
    return rt_multi_frame_create_rooted(comm, bvh, scene, cam, root_extra, 0, out);
}

int rt_multi_frame_create_rooted(rt_comm* comm, rt_bvh* bvh, const rt_scene* scene, const rt_camera* cam, uint32_t root_extra, int root, rt_multi_frame** out)
{
    if (!comm || !bvh || !scene || !out) return RT_FAIL(comm ? comm->ctx : nullptr, "null argument");
    rt_ctx* ctx = comm->ctx;
    *out = nullptr;
    if (root < 0 || root >= comm->nranks) return RT_FAIL(ctx, "root rank outside the communicator");
    if (bvh->ctx != ctx || scene->ctx != ctx) return RT_FAIL(ctx, "the BVH and the scene must live on the communicator's context");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (root_extra > 0xFFFFu) return RT_FAIL(ctx, "root_extra too large");
    RT_ENTER(ctx);
    rt_multi_frame* f = new (std::nothrow) rt_multi_frame();
    if (!f) return RT_FAIL(ctx, "out of host memory");
    f->comm = comm; f->bvh = bvh; f->scene = scene; f->cam = *cam;
    const uint32_t N = (uint32_t)comm->nranks;
    f->root = root; f->vrank = (comm->rank - root + comm->nranks) % comm->nranks;
    f->tiling = RT_TILING(N, N > 1 ? root_extra : 0u);
    f->own_slots = rt_tiles_owned_pixels(cam, (uint32_t)f->vrank, f->tiling);
    f->root_slots = rt_tiles_owned_pixels(cam, 0, f->tiling);
    for (uint32_t r = 1; r < N; r++) { const uint32_t s = rt_tiles_owned_pixels(cam, r, f->tiling); if (s > f->peer_slots) f->peer_slots = s; }
    const uint32_t plane = f->vrank == 0 ? f->root_slots : f->peer_slots;        // plane stride of this rank's send buffers
    hipError_t e = hipMalloc(&f->d_contacts, (size_t)(f->own_slots ? f->own_slots : 1u) * sizeof(rt_contact));
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipMalloc((void**)&f->d_send[b], (size_t)(plane ? plane : 1u) * 3u * sizeof(float));
        if (e == hipSuccess) e = hipMemsetAsync(f->d_send[b], 0, (size_t)(plane ? plane : 1u) * 3u * sizeof(float), ctx->stream);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->traced[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->gathered[b], hipEventDisableTiming);
    }
    if (e == hipSuccess && f->vrank == 0) {
        e = hipMalloc((void**)&f->d_frame, (size_t)cam->resX * cam->resY * sizeof(float4));
        if (e == hipSuccess && N > 1) e = hipMalloc((void**)&f->d_peers, (size_t)(N - 1) * 3u * f->peer_slots * sizeof(float));
    }
    if (e != hipSuccess) { rt_multi_frame_destroy(f); (void)hipGetLastError(); return RT_FAIL(ctx, std::string("multi-GPU frame buffers: ") + hipGetErrorString(e)); }
    // one frame graph per send buffer: trace of this rank's tiles with the calibrated strategy + shading into the planes
    if (f->own_slots) {
        for (int b = 0; b < 2; b++)
            if (rt_frame_graph_create(bvh, scene, cam, (uint32_t)f->vrank, f->tiling, RT_FRAME_TILED_PACKED, f->d_contacts, f->d_send[b], plane,
                                      &f->graph[b])) { rt_multi_frame_destroy(f); return RT_ERROR; }
    }
    *out = f;
    return RT_SUCCESS;
}

// One frame of every given rank: n = 1 in a job with one process per GPU, n = the group size in a single-process group (all
// of its ranks, in rank order).  Asynchronous: the trace runs on each context's stream, the gather and the untile on each
// communicator's stream; rt_multi_frame_wait waits for both.
int rt_multi_frame_render(rt_multi_frame** frames, int n)
{
    if (!frames || n < 1 || !frames[0]) return RT_FAIL(nullptr, "invalid argument");
    rt_ctx* ctx0 = frames[0]->comm->ctx;
    const rt_comm* c0 = frames[0]->comm;
    if (!c0->group.empty() ? n != c0->nranks : n != 1) return RT_FAIL(ctx0, "a single-process group renders all of its ranks in one call, a multi-process rank alone");
    for (int i = 0; i < n; i++) {
        if (frames[i] && frames[i]->comm->broken) return RT_FAIL(ctx0, "this communicator is broken: an earlier gather failed half way");
        if (!frames[i] || frames[i]->comm->nranks != c0->nranks || frames[i]->frames != frames[0]->frames) return RT_FAIL(ctx0, "frames of different groups or out of step");
        if (frames[i]->root != frames[0]->root) return RT_FAIL(ctx0, "the frames of one render call gather into one root");
        if (!c0->group.empty() && frames[i]->comm != c0->group[(size_t)i]) return RT_FAIL(ctx0, "frames must be given in rank order");
    }
    const int b = (int)(frames[0]->frames & 1u);
    const uint32_t N = (uint32_t)c0->nranks;
    // 1. every rank: trace + shade its tiles into send[b] (once the gather that last read send[b] is through)
    for (int i = 0; i < n; i++) {
        rt_multi_frame* f = frames[i];
        rt_ctx* ctx = f->comm->ctx;
        RT_ENTER(ctx);
        if (f->timing) harvest_timing(f, b);
        if (f->gathered_valid[b]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, f->gathered[b], 0));
        if (f->timing) RT_HIP(ctx, hipEventRecord(f->tev[b][0], ctx->stream));
        if (f->graph[b] && rt_frame_graph_launch(f->graph[b])) return RT_ERROR;
        if (f->timing) RT_HIP(ctx, hipEventRecord(f->tev[b][1], ctx->stream));
        RT_HIP(ctx, hipEventRecord(f->traced[b], ctx->stream));
        RT_HIP(ctx, hipStreamWaitEvent(f->comm->stream, f->traced[b], 0));
        if (f->timing) RT_HIP(ctx, hipEventRecord(f->tev[b][2], f->comm->stream));
    }
    // 2. the one collective: the peers' planes to the root
    if (N > 1 && c0->nccl) {
        RT_NCCL(ctx0, g_rccl.GroupStart());
        // a failing Send / Recv must not leave the group open on this thread (the next RCCL call would hang or misbehave): close
        // it, mark the communicators broken -- this rank's frame counters have not advanced, the peers' may have -- and report
        // the FIRST error
        ncclResult_t bad = 0; const char* what = nullptr; rt_ctx* bad_ctx = ctx0;
        for (int i = 0; i < n && !bad; i++) {
            rt_multi_frame* f = frames[i];
            rt_comm* c = f->comm;
            const size_t count = (size_t)3u * f->peer_slots;
            (void)hipSetDevice(c->ctx->device);
            if (f->vrank == 0) {
                for (uint32_t r = 1; r < N && !bad; r++) {        // virtual rank r = communicator rank (r + root) mod N
                    bad = g_rccl.Recv(f->d_peers + (size_t)(r - 1u) * count, count, rtNcclFloat32, (int)((r + (uint32_t)f->root) % N), c->nccl, c->stream);
                    if (bad) { what = "ncclRecv"; bad_ctx = c->ctx; }
                }
            } else if (count) {
                bad = g_rccl.Send(f->d_send[b], count, rtNcclFloat32, f->root, c->nccl, c->stream);
                if (bad) { what = "ncclSend"; bad_ctx = c->ctx; }
            }
        }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (bad || end) {
            for (int i = 0; i < n; i++) frames[i]->comm->broken = true;
            return RT_FAIL(bad_ctx, std::string(bad ? what : "ncclGroupEnd") + ": " + g_rccl.GetErrorString(bad ? bad : end) +
                                    " (the gather was abandoned; this communicator renders no further frames)");
        }
    } else if (N > 1) {
        // contexts that share a device (one-GPU tests): the root's stream copies every peer's planes once that peer has traced
        rt_multi_frame* root = frames[frames[0]->root];
        rt_ctx* rctx = root->comm->ctx;
        RT_ENTER(rctx);
        const size_t count = (size_t)3u * root->peer_slots;
        for (int i = 0; i < n; i++) {
            if (frames[i] == root) continue;
            RT_HIP(rctx, hipStreamWaitEvent(root->comm->stream, frames[i]->traced[b], 0));
            if (count) RT_HIP(rctx, hipMemcpyAsync(root->d_peers + (size_t)(frames[i]->vrank - 1) * count, frames[i]->d_send[b], count * sizeof(float), hipMemcpyDeviceToDevice, root->comm->stream));
        }
    }
    // 3. root: planes -> pixel order; everybody: send[b] is free again after this point of the communicator's stream
    for (int i = 0; i < n; i++) {
        rt_multi_frame* f = frames[i];
        rt_comm* c = f->comm;
        RT_ENTER(c->ctx);
        if (f->vrank == 0) {
            launch_untile_rgba_split(c->stream, f->cam, f->tiling, f->d_send[b], f->root_slots, f->d_peers, f->peer_slots, (uint64_t)3u * f->peer_slots, f->d_frame);
            RT_HIP(c->ctx, hipGetLastError());
        }
        // (copy transport: a peer's buffer is read by the ROOT's stream, so that is where its event goes)
        hipStream_t reader = (c->by_copies && f->vrank != 0) ? frames[f->root]->comm->stream : c->stream;
        RT_HIP(c->ctx, hipEventRecord(f->gathered[b], reader));
        if (f->timing) { RT_HIP(c->ctx, hipEventRecord(f->tev[b][3], reader)); f->tev_pending[b] = true; }
        f->gathered_valid[b] = true;
        f->frames++;
    }
    return RT_SUCCESS;
}

int rt_multi_frame_wait(rt_multi_frame* f)
{
    if (!f) return RT_FAIL(nullptr, "frame is null");
    rt_ctx* ctx = f->comm->ctx;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(f->comm->stream));
    return rt_ctx_synchronize(ctx);               // + the trace kernels' error word
}

void* rt_multi_frame_pixels(const rt_multi_frame* f) { return f ? (void*)f->d_frame : nullptr; }

int rt_multi_frame_set_timing(rt_multi_frame* f, int enable)
{
    if (!f) return RT_FAIL(nullptr, "frame is null");
    rt_ctx* ctx = f->comm->ctx;
    RT_ENTER(ctx);
    for (int b = 0; b < 2; b++) { harvest_timing(f, b); }
    if (enable)
        for (int b = 0; b < 2; b++) for (int k = 0; k < 4; k++) if (!f->tev[b][k]) RT_HIP(ctx, hipEventCreate(&f->tev[b][k]));
    f->timing = enable != 0;
    f->trace_ms_sum = f->gather_ms_sum = 0.0; f->timed_frames = 0;
    return RT_SUCCESS;
}

int rt_multi_frame_times(rt_multi_frame* f, double* trace_ms, double* gather_ms, uint64_t* frames)
{
    if (!f) return RT_FAIL(nullptr, "frame is null");
    RT_ENTER(f->comm->ctx);
    for (int b = 0; b < 2; b++) harvest_timing(f, b);
    if (trace_ms) *trace_ms = f->trace_ms_sum;
    if (gather_ms) *gather_ms = f->gather_ms_sum;
    if (frames) *frames = f->timed_frames;
    f->trace_ms_sum = f->gather_ms_sum = 0.0; f->timed_frames = 0;
    return RT_SUCCESS;
}
int rt_multi_frame_root(const rt_multi_frame* f) { return f ? f->root : -1; }

// The camera of a tiled frame moves like the camera of any captured frame (rt_frame_graph_set_camera): both send buffers' graphs get
// the new camera as new kernel arguments; every rank passes the same camera (same resolution as at creation).
int rt_multi_frame_set_camera(rt_multi_frame* f, const rt_camera* cam)
{
    if (!f) return RT_FAIL(nullptr, "frame is null");
    rt_ctx* ctx = f->comm->ctx;
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (cam->resX != f->cam.resX || cam->resY != f->cam.resY) return RT_FAIL(ctx, "multi-GPU frame: the resolution is part of the frame; create a new one");
    for (int b = 0; b < 2; b++)
        if (f->graph[b] && rt_frame_graph_set_camera(f->graph[b], cam)) return RT_ERROR;
    f->cam = *cam;
    return RT_SUCCESS;
}

int rt_multi_frame_layout(const rt_multi_frame* f, uint32_t* tiling, uint32_t* own_slots, uint32_t* root_slots, uint32_t* peer_slots)
{
    if (!f) return RT_FAIL(nullptr, "frame is null");
    if (tiling) *tiling = f->tiling;
    if (own_slots) *own_slots = f->own_slots;
    if (root_slots) *root_slots = f->root_slots;
    if (peer_slots) *peer_slots = f->peer_slots;
    return RT_SUCCESS;
}

}  // extern "C"
