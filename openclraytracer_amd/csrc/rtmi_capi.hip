// rtmi_capi.hip -- the C ABI of librtmi.so (include/rtmi.h), part 1: context, buffers, scene upload, shading / tiling entry points,
// primitives, measurement helpers.  (rtmi_bvh.hip: the BVH manager's build; rtmi_trace_policy.hip: which kernels trace a frame;
// rtmi_frame_graph.hip: captured frames; rtmi_rays.hip: the ray-buffer entry; rtmi_tlg_capi.hip, rtmi_multi.hip.)
// Mirrors the call sequence of the reference managers (BVHManager.cpp, Scene.cpp, CLBuffer.cpp)
// but enqueues hand-written HIP kernels on one HIP stream instead of JIT-compiled OpenCL.
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

static thread_local std::string g_null_ctx_error;

int rt_fail(const rt_ctx* ctx, const char* file, int line, const char* func, const std::string& msg)
{
    std::ostringstream os;
    // same fields as Common::Errata (Errata.h:58-110): file(line) function message
    os << file << '(' << line << ") " << func << ' ' << msg;
    if (ctx) ctx->last_error = os.str(); else g_null_ctx_error = os.str();
    return RT_ERROR;
}

static void free_ctx_buffers(rt_ctx* ctx)
{
    if (ctx->d_trace_ctrl) (void)hipFree(ctx->d_trace_ctrl);
    if (ctx->d_long_list) (void)hipFree(ctx->d_long_list);
    if (ctx->d_keys) (void)hipFree(ctx->d_keys);
    if (ctx->d_split_list) (void)hipFree(ctx->d_split_list);
    if (ctx->d_clock) (void)hipFree(ctx->d_clock);
    for (uint32_t i = 0; i < ctx->n_tile_tables; i++) if (ctx->tile_tables[i].d) (void)hipFree(ctx->tile_tables[i].d);
    if (ctx->d_task_list) (void)hipFree(ctx->d_task_list);
    ctx->d_task_list = nullptr;
    ctx->d_trace_ctrl = nullptr; ctx->d_long_list = nullptr; ctx->long_cap = 0;
    ctx->d_keys = nullptr; ctx->keys_cap = 0; ctx->d_split_list = nullptr;
}

// records for rays deferred from the lane-per-ray phase to the wave-per-ray phase: grow-only,
// one per work item up to RT_LONG_LIST_MAX (beyond that rays simply finish in the first phase)
int ensure_long_list(rt_ctx* ctx, uint64_t work_items)
{
    uint64_t limit = ctx->long_list_limit ? ctx->long_list_limit : RT_LONG_LIST_MAX;
    uint64_t want = work_items < limit ? work_items : limit;
    if (want < 1024 && !ctx->long_list_limit) want = 1024;
    if (want > ctx->long_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_long_list) { (void)hipFree(ctx->d_long_list); ctx->d_long_list = nullptr; ctx->long_cap = 0; }
        if (ctx->d_split_list) { (void)hipFree(ctx->d_split_list); ctx->d_split_list = nullptr; }
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_long_list, want * trace_long_record_bytes()));
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_split_list, want * sizeof(uint32_t)));
        ctx->long_cap = (uint32_t)want;
        ctx->scratch_generation++;                 // frame graphs created before this hold the freed addresses
        // test knob: what recycled device memory looks like
        if (ctx->poison & 1) RT_HIP(ctx, hipMemsetAsync(ctx->d_long_list, 0xCD, want * trace_long_record_bytes(), ctx->stream));
        if (ctx->poison & 2) RT_HIP(ctx, hipMemsetAsync(ctx->d_split_list, 0xCD, want * sizeof(uint32_t), ctx->stream));
    }
    // queue capacity by frame size: ~8 entries per 8x8 block over the 256 queues (a 4K frame of the atrium publishes 2-3 per block)
    if (ctx->task_cap_auto) {
        uint64_t want_cap = work_items / 64u * 8u / 256u;
        want_cap = want_cap < 64 ? 64 : want_cap > 4096 ? 4096 : want_cap;
        if (want_cap > ctx->task_cap) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_task_list) { (void)hipFree(ctx->d_task_list); ctx->d_task_list = nullptr; }
            ctx->task_cap = (uint32_t)want_cap;
        }
    }
    if (!ctx->d_task_list && ctx->task_cap) {
        // sub-tree tasks the waves of the four-wide packet kernel share (16 bytes each); a full queue only means less sharing
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_task_list, trace_task_bytes(ctx->task_cap)));
        ctx->scratch_generation++;
    }
    if (work_items > ctx->keys_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_keys) { (void)hipFree(ctx->d_keys); ctx->d_keys = nullptr; ctx->keys_cap = 0; }
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_keys, work_items * sizeof(unsigned long long)));
        ctx->keys_cap = work_items;
        ctx->scratch_generation++;
        if (ctx->poison & 4) RT_HIP(ctx, hipMemsetAsync(ctx->d_keys, 0xCD, work_items * sizeof(unsigned long long), ctx->stream));
    }
    return RT_SUCCESS;
}

// The trace kernels report a failed frame (deferred list full on a forced hand-off, pool / stack overflow, bad record)
// through one word of its control block; every synchronising call looks at it once per launch batch.
int check_trace_error(rt_ctx* ctx)
{
    if (!ctx->trace_launched) return RT_SUCCESS;
    ctx->trace_launched = false;
    uint32_t err = 0;
    void* word = (char*)ctx->d_trace_ctrl + trace_ctrl_error_offset();
    RT_HIP(ctx, hipMemcpy(&err, word, 4, hipMemcpyDeviceToHost));
    if (!err) return RT_SUCCESS;
    // the word is sticky on the device (kernels OR a bit per failure, launches do not clear it): every invalid frame since the
    // last check is reported here, once
    RT_HIP(ctx, hipMemset(word, 0, 4));
    std::string msg = "trace kernel:";
    if (err & (1u << 1)) msg += " deferred-ray list full while a full traversal stack had to hand off work;";
    if (err & (1u << 2)) msg += " node pool of the wave-per-ray phase overflowed;";
    if (err & (1u << 3)) msg += " packet stack overflowed its spill area;";
    if (err & (1u << 4)) msg += " a deferred-ray record does not describe a task of this launch;";
    if (err & (1u << 5)) msg += " no free spill area for a packet stack;";
    if (err & ~0x3Eu) msg += " unknown failure bits;";
    return RT_FAIL(ctx, msg + " frame invalid");
}

extern "C" {

int rt_abi_version(void) { return RTMI_ABI_VERSION; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_null_ctx_error.c_str(); }

int rt_ctx_create(int device_ordinal, rt_ctx** out)
{
    if (!out) return RT_FAIL(nullptr, "out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return RT_FAIL(nullptr, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count) return RT_FAIL(nullptr, "device ordinal out of range");
    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_FAIL(nullptr, "out of host memory");
    ctx->device = device_ordinal;
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device_ordinal)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc(&ctx->d_trace_ctrl, trace_ctrl_bytes())) != hipSuccess ||
        (e = hipMemset(ctx->d_trace_ctrl, 0, trace_ctrl_bytes())) != hipSuccess) {      // launches clear the counters only: the error word starts clean here
        RT_FAIL(nullptr, std::string("context creation failed: ") + hipGetErrorString(e));
        free_ctx_buffers(ctx);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return RT_ERROR;
    }
    if (std::string(ctx->prop.gcnArchName).rfind("gfx950", 0) != 0) {
        RT_FAIL(nullptr, std::string("librtmi is built for gfx950 only, device is ") + ctx->prop.gcnArchName);
        free_ctx_buffers(ctx); (void)hipStreamDestroy(ctx->own_stream); delete ctx;
        return RT_ERROR;
    }
    ctx->stream = ctx->own_stream;
    {
        // Which HIP runtime serves the process: librtmi.so is built against the system's (HIP_VERSION at compile time); a process that
        // imported torch first runs on the copy torch bundles, an older ROCm.  Measured in round 3 under that copy: an executable graph
        // whose kernel-node arguments were re-set (hipGraphExecKernelNodeSetParams: a camera that moves) replays three times slower, and
        // five streams of such graphs stall.  Frame graphs therefore never re-set node arguments there (rtmi_frame_graph.hip: sequences of
        // kernels only are issued as plain launches with refreshed argument copies under every runtime; any other sequence falls back to the
        // eager calls for a moved camera when the runtime is the older one).  RTMI_GRAPH_SETPARAMS=1 / 0 overrides the detection.
        int rv = 0;
        if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
        ctx->older_runtime = rv != 0 && rv < HIP_VERSION;
        if (const char* env = std::getenv("RTMI_GRAPH_SETPARAMS")) ctx->older_runtime = env[0] == '0';
    }
    // (failures are not fatal: the object is then loaded where it is first used)
    (void)rt_preload_trace(); (void)rt_preload_build(); (void)rt_preload_sort_scan(); (void)rt_preload_shade(); (void)rt_preload_tlg(); (void)rt_preload_secondary();
    if (const char* env = std::getenv("RTMI_DEFER_BUDGET")) ctx->defer_budget = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_LONG_LIST_MAX")) ctx->long_list_limit = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_POISON")) ctx->poison = (int)std::strtol(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_TRACE_DIAG")) ctx->trace_diagnostics = env[0] == '1';
    if (const char* env = std::getenv("RTMI_CALIBRATION_LOG")) ctx->calibration_log = env[0] == '1';
    if (const char* env = std::getenv("RTMI_RECALIBRATE")) ctx->recalibrate = env[0] == '1';
    if (const char* env = std::getenv("RTMI_REFIT")) ctx->refit_choice = env[0] == 'r' ? 1 : env[0] == 'c' ? 2 : env[0] == 's' ? 3 : 0;
    if (const char* env = std::getenv("RTMI_SEARCH_TREE")) ctx->no_search_tree = env[0] == '0';
    if (const char* env = std::getenv("RTMI_PACKET")) { ctx->force_packet = env[0] == '1' || env[0] == '4'; ctx->force_packet4 = env[0] == '4'; ctx->no_packet = env[0] == '0'; }
    if (const char* env = std::getenv("RTMI_PACKET4")) ctx->no_packet4 = env[0] == '0';
    ctx->task_cap = 64; ctx->task_cap_auto = true;     // per queue, 256 queues; grows with the frame (ensure_long_list)
    if (const char* env = std::getenv("RTMI_TASK_CAP")) { ctx->task_cap = (uint32_t)std::strtoul(env, nullptr, 10); ctx->task_cap_auto = false; }   // 0: no sharing
    if (const char* env = std::getenv("RTMI_FUSED_FRAMES")) ctx->no_fused_frames = env[0] == '0';
    if (const char* env = std::getenv("RTMI_BLOCK_ORDER")) ctx->no_block_order = env[0] == '0';
    if (const char* env = std::getenv("RTMI_BLOCK_ORDER_IN_FLIGHT")) ctx->order_in_flight = env[0] != '0';
    if (const char* env = std::getenv("RTMI_TRACE_WORKGROUP")) { const uint32_t v = (uint32_t)std::strtoul(env, nullptr, 10); if (v == 64u || v == 128u || v == 256u) ctx->trace_workgroup_waves = v / 64u; }
    if (const char* env = std::getenv("RTMI_SEARCH_TREE_BOXES")) ctx->search_tree_reference_boxes = std::strcmp(env, "reference") == 0;
    if (const char* env = std::getenv("RTMI_BLOCK_ORDER_GROUP")) { const uint32_t g = (uint32_t)std::strtoul(env, nullptr, 10); if (g && g <= 64u && !(g & (g - 1u))) { ctx->block_order_group_log = 0; while ((1u << ctx->block_order_group_log) < g) ctx->block_order_group_log++; } }
    if (const char* env = std::getenv("RTMI_SHARE_MIN_BUDGET")) ctx->share_min_budget = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_TASK_HELPERS")) ctx->task_helper_blocks = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_QUAD_BY_AREA")) ctx->quad_by_area = env[0] != '0';
    ctx->no_octant_records = std::getenv("RTMI_NO_OCTANT_RECORDS") != nullptr;
    if (const char* env = std::getenv("RTMI_BUILD_GRAPH")) ctx->no_build_graph = env[0] == '0';
    if (const char* env = std::getenv("RTMI_GRAPH_DIRECT")) ctx->no_graph_direct = env[0] == '0';
    if (const char* env = std::getenv("RTMI_GRAPH_DEBUG")) ctx->graph_debug = env[0] == '1' ? (std::getenv("RTMI_GRAPH_DEBUG_VERBOSE") ? 2 : 1) : 0;
    if (const char* env = std::getenv("RTMI_RAYS")) ctx->rays_forced = env;
    if (const char* env = std::getenv("RTMI_TLG_REBUILD_ON_PREVIOUS_SIZES")) ctx->tlg_no_rebuild_on_previous = env[0] == '0';
    if (const char* env = std::getenv("RTMI_TLG_TRICAM")) ctx->tlg_no_tricam = env[0] == '0';
    if (const char* env = std::getenv("RTMI_TLG_WALK_RECORDS")) ctx->tlg_no_walk_records = env[0] == '0';
    if (const char* env = std::getenv("RTMI_CALIBRATION_DEADLINE")) ctx->no_calibration_deadline = env[0] == '0';
    if (const char* env = std::getenv("RTMI_IN_FLIGHT_SHARE_HANDICAP")) { const float v = (float)std::atof(env); if (v >= 1.0f) ctx->in_flight_share_handicap = v; }
    if (const char* env = std::getenv("RTMI_CALIBRATE_ALL")) ctx->calibrate_all = env[0] == '1';
    if (const char* env = std::getenv("RTMI_SEARCH_KEYS")) ctx->search_keys_ten_per_axis = std::strcmp(env, "cubic") == 0;
    if (const char* env = std::getenv("RTMI_LONG_BATCH")) { const long v = std::strtol(env, nullptr, 10); if (v == 1 || v == 2 || v == 4) ctx->long_batch = (uint32_t)v; }
    *out = ctx;
    return RT_SUCCESS;
}

void rt_ctx_destroy(rt_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_ctx_buffers(ctx);
    for (hipEvent_t e : ctx->trace_events) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RT_SUCCESS;
}

void* rt_ctx_get_stream(const rt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_ctx_synchronize(rt_ctx* ctx)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_trace_error(ctx);
}

int rt_ctx_device_info(const rt_ctx* ctx, int* compute_units, int* wave_size, uint64_t* lds_bytes, char* name,
                       size_t name_len)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (wave_size) *wave_size = ctx->prop.warpSize;
    if (lds_bytes) *lds_bytes = ctx->prop.maxSharedMemoryPerMultiProcessor;
    if (name && name_len) { std::strncpy(name, ctx->prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------
int rt_buf_alloc(rt_ctx* ctx, uint64_t bytes, void** d_out)
{
    if (!ctx || !d_out) return RT_FAIL(ctx, "null argument");
    *d_out = nullptr;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipMalloc(d_out, bytes ? bytes : 1));
    return RT_SUCCESS;
}

int rt_buf_free(rt_ctx* ctx, void* d_ptr)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (!d_ptr) return RT_SUCCESS;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipFree(d_ptr));
    note_ray_buffer(ctx, d_ptr, RT_RAYKIND_FOREIGN);          // the address may come back as another buffer
    return RT_SUCCESS;
}

int rt_buf_upload(rt_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !h_src))) return RT_FAIL(ctx, "null argument");
    if (!bytes) return RT_SUCCESS;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // the host buffer may be reused on return
    note_ray_buffer(ctx, d_dst, RT_RAYKIND_FOREIGN);    // (if it was a ray buffer of the library's generators: the caller's bytes now)
    return RT_SUCCESS;
}

int rt_buf_download(rt_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!h_dst || !d_src))) return RT_FAIL(ctx, "null argument");
    if (!bytes) return RT_SUCCESS;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_trace_error(ctx);
}

__global__ void k_fill_u32(uint32_t* dst, uint32_t pattern, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = pattern;
}

int rt_buf_fill_u32(rt_ctx* ctx, void* d_dst, uint32_t pattern, uint64_t count)
{
    if (!ctx || (count && !d_dst)) return RT_FAIL(ctx, "null argument");
    if (!count) return RT_SUCCESS;
    RT_ENTER(ctx);
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (uint32_t*)d_dst, pattern, count);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------------------
static uint64_t rd64(const unsigned char* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

int rt_scene_upload(rt_ctx* ctx, const void* h_scene_buffer, uint64_t bytes, rt_scene** out)
{
    return rt_scene_upload_indexed(ctx, h_scene_buffer, bytes, 2, out);
}

int rt_scene_upload_indexed(rt_ctx* ctx, const void* h_scene_buffer, uint64_t bytes, int index_bytes, rt_scene** out)
{
    if (!ctx || !h_scene_buffer || !out) return RT_FAIL(ctx, "null argument");
    *out = nullptr;
    if (index_bytes != 2 && index_bytes != 4) return RT_FAIL(ctx, "index width must be 2 or 4 bytes");
    const uint64_t IB = (uint64_t)index_bytes;
    const unsigned char* s = (const unsigned char*)h_scene_buffer;
    // SceneHeader, SceneBufferParser.h:55-66
    if (bytes < 96) return RT_FAIL(ctx, "scene buffer smaller than a SceneHeader");
    uint64_t totalDataSize = rd64(s), nLights = rd64(s + 16), nSpheres = rd64(s + 24), nMaterials = rd64(s + 32),
             nModels = rd64(s + 48), nTris = rd64(s + 56);
    if (totalDataSize > bytes) return RT_FAIL(ctx, "SceneHeader.totalDataSize exceeds the buffer");
    if (nLights > bytes || nSpheres > bytes || nMaterials > bytes) return RT_FAIL(ctx, "corrupt scene header counts");
    uint64_t off = 96 + 16 * nLights + 16 * nSpheres + 96 * nMaterials;        // MODEL_BUFFER_PTR, :69-74
    if (off > bytes) return RT_FAIL(ctx, "scene header counts exceed the buffer");
    // record and leaf ids carry a ray slot in their top four bits inside the wave-per-ray kernel (2 (n-1) records < 2^28)
    if (nTris > (1ull << 27)) return RT_FAIL(ctx, "too many triangles (more than 2^27)");

    rt_scene* sc = new (std::nothrow) rt_scene();
    if (!sc) return RT_FAIL(ctx, "out of host memory");
    sc->ctx = ctx;
    uint64_t tri_acc = 0;
    for (uint64_t m = 0; m < nModels; m++) {
        if (off + 64 > bytes) { delete sc; return RT_FAIL(ctx, "model header outside the buffer"); }
        uint64_t modelSize = rd64(s + off), nSub = rd64(s + off + 8);           // ModelHeader, :129-136
        if (modelSize < 64 || off + modelSize > bytes) { delete sc; return RT_FAIL(ctx, "model size outside the buffer"); }
        uint64_t moff = off + 64;
        for (uint64_t k = 0; k < nSub; k++) {
            if (moff + 48 > off + modelSize) { delete sc; return RT_FAIL(ctx, "mesh header outside its model"); }
            uint64_t meshSize = rd64(s + moff), mt = rd64(s + moff + 8), nv = rd64(s + moff + 16), ni = rd64(s + moff + 24),
                     mat = rd64(s + moff + 32);                                  // MeshHeader, :141-149
            if (meshSize < 48 + 16 * nv + IB * ni || moff + meshSize > off + modelSize || mt * 3 > ni) {
                delete sc; return RT_FAIL(ctx, "mesh sizes inconsistent");
            }
            for (uint64_t v = 0; v < nv && !sc->has_nan; v++) {          // (a NaN coordinate makes boxes NaN: the refit by spans then keeps its fall-back)
                float xyz[3]; std::memcpy(xyz, s + moff + 48 + 16 * v, 12);
                if (xyz[0] != xyz[0] || xyz[1] != xyz[1] || xyz[2] != xyz[2]) sc->has_nan = true;
            }
            const unsigned char* ib = s + moff + 48 + 16 * nv;
            for (uint64_t i = 0; i < mt * 3; i++) {
                uint32_t v = 0; std::memcpy(&v, ib + IB * i, IB);
                if (v >= nv) { delete sc; return RT_FAIL(ctx, "vertex index out of range"); }
            }
            MeshEntry e{};
            e.byte_offset = moff; e.first_triangle = (uint32_t)tri_acc; e.triangle_count = (uint32_t)mt;
            e.vertex_count = (uint32_t)nv; e.material_index = (uint32_t)mat;
            e.model_index = (uint32_t)m; e.submesh_index = (uint32_t)k;
            sc->h_meshes.push_back(e);
            tri_acc += mt;
            moff += meshSize;
        }
        off += modelSize;
    }
    if (tri_acc != nTris) { delete sc; return RT_FAIL(ctx, "SceneHeader.totalNumberOfTriangles does not match the meshes"); }

    hipError_t e = hipSetDevice(ctx->device);
    (void)hipGetLastError();                       // see RT_ENTER
    size_t tri_bytes = (size_t)nTris * (6 * sizeof(float4) + sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc(&sc->d_bytes, bytes);
    if (e == hipSuccess) e = hipMalloc(&sc->d_meshes, sizeof(MeshEntry) * (sc->h_meshes.size() + 1));
    if (e == hipSuccess) e = hipMalloc(&sc->d_tri, tri_bytes ? tri_bytes : 16);
    if (e == hipSuccess) e = hipMemcpyAsync(sc->d_bytes, s, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && !sc->h_meshes.empty())
        e = hipMemcpyAsync(sc->d_meshes, sc->h_meshes.data(), sizeof(MeshEntry) * sc->h_meshes.size(),
                           hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        RT_FAIL(ctx, std::string("scene upload failed: ") + hipGetErrorString(e));
        rt_scene_destroy(sc);
        return RT_ERROR;
    }
    DeviceScene& d = sc->d;
    d.bytes = (const unsigned char*)sc->d_bytes; d.nbytes = bytes;
    d.n_tris = (uint32_t)nTris; d.n_meshes = (uint32_t)sc->h_meshes.size();
    d.meshes = (const MeshEntry*)sc->d_meshes;
    float4* base = (float4*)sc->d_tri;
    d.v0 = base; d.v1 = base + nTris; d.v2 = base + 2 * nTris; d.e1 = base + 3 * nTris; d.e2 = base + 4 * nTris;
    d.nrm = base + 5 * nTris; d.ref = (uint4*)(base + 6 * nTris);
    d.index_bytes = (uint32_t)index_bytes;
    d.n_lights = (uint32_t)nLights; d.n_spheres = (uint32_t)nSpheres; d.n_materials = (uint32_t)nMaterials;
    std::memcpy(d.bb_lo, s + 64, 12);
    std::memcpy(d.bb_hi, s + 80, 12);
    std::memcpy(&sc->bb_w[0], s + 76, 4); std::memcpy(&sc->bb_w[1], s + 92, 4);
    launch_gather_triangles(ctx->stream, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        RT_FAIL(ctx, std::string("triangle gather failed: ") + hipGetErrorString(e));
        rt_scene_destroy(sc);
        return RT_ERROR;
    }
    *out = sc;
    return RT_SUCCESS;
}

void rt_scene_destroy(rt_scene* sc)
{
    if (!sc) return;
    if (sc->ctx) { (void)hipSetDevice(sc->ctx->device); (void)hipStreamSynchronize(sc->ctx->stream); }
    if (sc->d_bytes) (void)hipFree(sc->d_bytes);
    if (sc->d_meshes) (void)hipFree(sc->d_meshes);
    if (sc->d_tri) (void)hipFree(sc->d_tri);
    delete sc;
}

const void* rt_scene_device_ptr(const rt_scene* sc) { return sc ? sc->d_bytes : nullptr; }
uint64_t rt_scene_bytes(const rt_scene* sc) { return sc ? sc->d.nbytes : 0; }
uint32_t rt_scene_triangle_count(const rt_scene* sc) { return sc ? sc->d.n_tris : 0; }

int rt_device_synchronize(int device_ordinal)
{
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return RT_FAIL(nullptr, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e));
    return RT_SUCCESS;
}

int rt_check_camera(rt_ctx* ctx, const rt_camera* cam)
{
    if (!cam) return RT_FAIL(ctx, "camera is null");
    if (cam->resX == 0 || cam->resY == 0) return RT_FAIL(ctx, "camera resolution is zero");
    if ((uint64_t)cam->resX * cam->resY > 0x3FFFFFFFull) return RT_FAIL(ctx, "camera resolution too large");
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// shading + tiling
// ---------------------------------------------------------------------------------------
int rt_shade_blinn_phong(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const void* d_contacts,
                         uint32_t contact_count, void* d_rgba)
{
    if (!ctx || !scene) return RT_FAIL(ctx, "null argument");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (!d_rgba || (contact_count && !d_contacts)) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    if (contact_count == 0) {
        RT_HIP(ctx, hipMemsetAsync(d_rgba, 0, (size_t)cam->resX * cam->resY * sizeof(float4), ctx->stream));
        return RT_SUCCESS;
    }
    launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_contacts, contact_count, (float4*)d_rgba, false, 0, 1);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_shade_blinn_phong_frame(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const void* d_primary_contacts, void* d_rgba)
{
    if (!ctx || !scene) return RT_FAIL(ctx, "null argument");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (!d_rgba || !d_primary_contacts) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    // contact i belongs to pixel i, so every pixel is written by its own thread (a miss as zeros) and the clearing pass of
    // rt_shade_blinn_phong -- 33 MB of writes per 1080p frame -- is not needed
    launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_primary_contacts, cam->resX * cam->resY, (float4*)d_rgba, true, 0, 1);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

uint32_t rt_tiles_owned_pixels(const rt_camera* cam, uint32_t rank, uint32_t nranks)
{
    if (!cam || rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return 0;
    uint32_t total = rt_tiles_x(cam->resX) * rt_tiles_y(cam->resY);
    return rt_owned_tiles(total, rank, nranks) * RT_TILE * RT_TILE;
}

int rt_shade_blinn_phong_tiled(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint32_t rank, uint32_t nranks,
                               const void* d_contacts_compact, void* d_rgba_compact)
{
    if (!ctx || !scene) return RT_FAIL(ctx, "null argument");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    if (!d_contacts_compact || !d_rgba_compact) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    uint32_t count = rt_tiles_owned_pixels(cam, rank, nranks);
    launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_contacts_compact, count, (float4*)d_rgba_compact, true,
                 rank, nranks);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_untile_rgba(rt_ctx* ctx, const rt_camera* cam, uint32_t nranks, const void* d_rgba_gathered, uint32_t slots_per_rank,
                   void* d_rgba_frame)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || !d_rgba_gathered || !d_rgba_frame) return RT_FAIL(ctx, "invalid argument");
    if (slots_per_rank < rt_tiles_owned_pixels(cam, 0, nranks)) return RT_FAIL(ctx, "slots_per_rank too small");
    RT_ENTER(ctx);
    launch_untile_rgba(ctx->stream, *cam, nranks, (const float4*)d_rgba_gathered, slots_per_rank, (float4*)d_rgba_frame);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_shade_blinn_phong_tiled_packed(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint32_t rank, uint32_t nranks,
                                      const void* d_contacts_compact, void* d_rgb_planes, uint32_t slots_per_rank)
{
    if (!ctx || !scene) return RT_FAIL(ctx, "null argument");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || rank >= rt_nranks(nranks)) return RT_FAIL(ctx, "rank/nranks invalid");
    if (!d_contacts_compact || !d_rgb_planes) return RT_FAIL(ctx, "null buffer");
    uint32_t count = rt_tiles_owned_pixels(cam, rank, nranks);
    if (slots_per_rank < count || slots_per_rank == 0) return RT_FAIL(ctx, "slots_per_rank too small");
    RT_ENTER(ctx);
    launch_shade(ctx->stream, scene->d, *cam, (const rt_contact*)d_contacts_compact, count, (float4*)d_rgb_planes, true,
                 rank, nranks, slots_per_rank);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_untile_rgba_packed(rt_ctx* ctx, const rt_camera* cam, uint32_t nranks, const void* d_rgb_planes_gathered,
                          uint32_t slots_per_rank, uint64_t rank_stride_floats, void* d_rgba_frame)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    if (rt_nranks(nranks) == 0 || !d_rgb_planes_gathered || !d_rgba_frame) return RT_FAIL(ctx, "invalid argument");
    if (slots_per_rank < rt_tiles_owned_pixels(cam, 0, nranks)) return RT_FAIL(ctx, "slots_per_rank too small");
    if (rank_stride_floats == 0) rank_stride_floats = 3ull * slots_per_rank;
    if (rank_stride_floats < 3ull * slots_per_rank) return RT_FAIL(ctx, "rank stride smaller than one rank's planes");
    RT_ENTER(ctx);
    launch_untile_rgba_packed(ctx->stream, *cam, nranks, (const float*)d_rgb_planes_gathered, slots_per_rank, rank_stride_floats,
                              (float4*)d_rgba_frame);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// primitives
// ---------------------------------------------------------------------------------------
int rt_sort_pairs_bitonic(rt_ctx* ctx, void* d_pairs, uint32_t n_pow2)
{
    if (!ctx || !d_pairs) return RT_FAIL(ctx, "null argument");
    if (n_pow2 == 0 || (n_pow2 & (n_pow2 - 1))) return RT_FAIL(ctx, "item count must be a power of two");
    RT_ENTER(ctx);
    launch_bitonic_sort(ctx->stream, (uint2*)d_pairs, n_pow2);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

int rt_sort_pairs(rt_ctx* ctx, void* d_pairs, uint32_t n, int algorithm)
{
    if (algorithm == RT_SORT_BITONIC) return rt_sort_pairs_bitonic(ctx, d_pairs, n);
    if (!ctx || (n && !d_pairs)) return RT_FAIL(ctx, "null argument");
    if (algorithm != RT_SORT_RADIX) return RT_FAIL(ctx, "unknown sort algorithm");
    if (n < 2) return RT_SUCCESS;
    RT_ENTER(ctx);
    void* tmp = nullptr;
    const size_t pair_bytes = ((size_t)n * sizeof(uint2) + 255u) & ~(size_t)255u;
    RT_HIP(ctx, hipMalloc(&tmp, pair_bytes + radix_sort_scratch_words(n) * sizeof(uint32_t)));
    launch_radix_sort(ctx->stream, (uint2*)d_pairs, (uint2*)tmp, n, 32u, (uint32_t*)((char*)tmp + pair_bytes));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // the scratch is freed below
    (void)hipFree(tmp);
    if (e != hipSuccess) return RT_FAIL(ctx, std::string("radix sort failed: ") + hipGetErrorString(e));
    return RT_SUCCESS;
}

int rt_scan_inclusive_u32(rt_ctx* ctx, const void* d_in, void* d_out, uint32_t n)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (n == 0) return RT_SUCCESS;
    if (!d_in || !d_out) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    uint32_t* scratch = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&scratch, scan_scratch_words(n) * sizeof(uint32_t)));
    launch_scan_inclusive(ctx->stream, (const uint32_t*)d_in, (uint32_t*)d_out, n, scratch);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // scratch is freed below
    (void)hipFree(scratch);
    if (e != hipSuccess) return RT_FAIL(ctx, std::string("scan failed: ") + hipGetErrorString(e));
    return RT_SUCCESS;
}

int rt_generate_secondary_rays(rt_ctx* ctx, const rt_camera* cam, const void* d_rays_in, const void* d_contacts, uint32_t count,
                               int kind, const float target[3], float eps, void* d_rays_out, uint32_t* out_count)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (out_count) *out_count = 0;
    if (count == 0) return RT_SUCCESS;
    if (!d_contacts || !d_rays_out || !out_count) return RT_FAIL(ctx, "null argument");
    if (!cam && !d_rays_in) return RT_FAIL(ctx, "either the camera or the incoming rays are needed");
    if (kind != RT_RAYS_SHADOW && kind != RT_RAYS_REFLECT) return RT_FAIL(ctx, "unknown ray kind");
    if (kind == RT_RAYS_SHADOW && !target) return RT_FAIL(ctx, "shadow rays need a target");
    if (cam && rt_check_camera(ctx, cam)) return RT_ERROR;
    RT_ENTER(ctx);
    uint32_t* tmp = nullptr;
    const size_t words = 2 * (size_t)count + scan_scratch_words(count);
    RT_HIP(ctx, hipMalloc((void**)&tmp, words * sizeof(uint32_t)));
    uint32_t* flags = tmp; uint32_t* scan = tmp + count; uint32_t* scratch = tmp + 2 * (size_t)count;
    const float zero[3] = { 0.0f, 0.0f, 0.0f };
    launch_secondary_flags(ctx->stream, (const rt_contact*)d_contacts, count, flags);
    launch_scan_inclusive(ctx->stream, flags, scan, count, scratch);
    launch_secondary_emit(ctx->stream, cam, (const rt_ray*)d_rays_in, (const rt_contact*)d_contacts, count, scan, kind,
                          target ? target : zero, eps, (rt_ray*)d_rays_out);
    hipError_t e = hipGetLastError();
    uint32_t n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n, scan + (count - 1), 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return RT_FAIL(ctx, std::string("secondary ray generation failed: ") + hipGetErrorString(e));
    *out_count = n;
    note_ray_buffer(ctx, d_rays_out, kind == RT_RAYS_SHADOW ? RT_RAYKIND_SHADOW : RT_RAYKIND_REFLECT);
    return RT_SUCCESS;
}

int rt_generate_primary_rays(rt_ctx* ctx, const rt_camera* cam, void* d_rays_out)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (!cam || !d_rays_out) return RT_FAIL(ctx, "null argument");
    if (rt_check_camera(ctx, cam)) return RT_ERROR;
    RT_ENTER(ctx);
    launch_primary_rays(ctx->stream, cam, (rt_ray*)d_rays_out);
    RT_HIP(ctx, hipGetLastError());
    note_ray_buffer(ctx, d_rays_out, RT_RAYKIND_PRIMARY);
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// per-kernel timing of the trace launches (HIP events on the stream the kernels run on)
// ---------------------------------------------------------------------------------------
static void drop_trace_events(rt_ctx* ctx)
{
    for (hipEvent_t e : ctx->trace_events) if (e) (void)hipEventDestroy(e);
    ctx->trace_events.clear();
}

int rt_ctx_set_frames_in_flight(rt_ctx* ctx, uint32_t n)
{
    if (!ctx) return RT_ERROR;
    ctx->frames_in_flight = n ? n : 1u;
    return RT_SUCCESS;
}

int rt_ctx_set_profiling(rt_ctx* ctx, int enable)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_trace_events(ctx);
    ctx->profiling = enable != 0;
    ctx->profiling_tag = enable == 2 ? 2 : 0;
    return RT_SUCCESS;
}

int rt_ctx_last_trace_counters(rt_ctx* ctx, uint32_t* out, uint32_t n)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    if (!out && n) return RT_FAIL(ctx, "counter array is null");
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t off[RT_TRACE_COUNTERS];
    trace_ctrl_counter_offsets(off);
    for (uint32_t k = 0; k < n; k++) {
        out[k] = 0;
        if (k < RT_TRACE_COUNTERS) RT_HIP(ctx, hipMemcpy(&out[k], (const char*)ctx->d_trace_ctrl + off[k], 4, hipMemcpyDeviceToHost));
    }
    return RT_SUCCESS;
}

int rt_ctx_trace_kernel_times(rt_ctx* ctx, uint64_t* launches, double* lane_phase_ms, double* wave_phase_ms)
{
    if (!ctx) return RT_FAIL(nullptr, "ctx is null");
    RT_ENTER(ctx);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n = 0; double a = 0.0, b = 0.0;
    for (size_t i = 0; i + 2 < ctx->trace_events.size(); i += 3) {
        hipEvent_t* e = &ctx->trace_events[i];
        float m0 = 0.f, m1 = 0.f;
        if (!e[0] || !e[1] || !e[2]) continue;
        if (hipEventElapsedTime(&m0, e[0], e[1]) != hipSuccess || hipEventElapsedTime(&m1, e[1], e[2]) != hipSuccess) continue;
        a += m0; b += m1; n++;
    }
    drop_trace_events(ctx);
    if (launches) *launches = n;
    if (lane_phase_ms) *lane_phase_ms = a;
    if (wave_phase_ms) *wave_phase_ms = b;
    return RT_SUCCESS;
}

// ---------------------------------------------------------------------------------------
// timers
// ---------------------------------------------------------------------------------------
struct rt_timer { hipEvent_t start, stop; };

int rt_timer_create(rt_ctx* ctx, void** timer)
{
    if (!ctx || !timer) return RT_FAIL(ctx, "null argument");
    RT_ENTER(ctx);
    rt_timer* t = new (std::nothrow) rt_timer();
    if (!t) return RT_FAIL(ctx, "out of host memory");
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
        delete t;
        return RT_FAIL(ctx, "hipEventCreate failed");
    }
    *timer = t;
    return RT_SUCCESS;
}
int rt_timer_start(rt_ctx* ctx, void* timer)
{
    if (!ctx || !timer) return RT_FAIL(ctx, "null argument");
    RT_HIP(ctx, hipEventRecord(((rt_timer*)timer)->start, ctx->stream));
    return RT_SUCCESS;
}
int rt_timer_stop(rt_ctx* ctx, void* timer)
{
    if (!ctx || !timer) return RT_FAIL(ctx, "null argument");
    RT_HIP(ctx, hipEventRecord(((rt_timer*)timer)->stop, ctx->stream));
    return RT_SUCCESS;
}
int rt_timer_elapsed_ms(rt_ctx* ctx, void* timer, float* ms)
{
    if (!ctx || !timer || !ms) return RT_FAIL(ctx, "null argument");
    rt_timer* t = (rt_timer*)timer;
    RT_HIP(ctx, hipEventSynchronize(t->stop));
    RT_HIP(ctx, hipEventElapsedTime(ms, t->start, t->stop));
    return RT_SUCCESS;
}
void rt_timer_destroy(rt_ctx* ctx, void* timer)
{
    if (!timer) return;
    rt_timer* t = (rt_timer*)timer;
    (void)hipEventDestroy(t->start); (void)hipEventDestroy(t->stop);
    delete t;
}

}  // extern "C"
