// rtmi_capi.hip -- the C ABI of librtmi.so (include/rtmi.h): host-side orchestration only.
// Mirrors the call sequence of the reference managers (BVHManager.cpp, Scene.cpp, CLBuffer.cpp)
// but enqueues hand-written HIP kernels on one HIP stream instead of JIT-compiled OpenCL.
#include "rtmi_internal.h"
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
static int ensure_long_list(rt_ctx* ctx, uint64_t work_items)
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
static int check_trace_error(rt_ctx* ctx)
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
    // (failures are not fatal: the object is then loaded where it is first used)
    (void)rt_preload_trace(); (void)rt_preload_build(); (void)rt_preload_sort_scan(); (void)rt_preload_shade(); (void)rt_preload_tlg(); (void)rt_preload_secondary();
    if (const char* env = std::getenv("RTMI_DEFER_BUDGET")) ctx->defer_budget = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_LONG_LIST_MAX")) ctx->long_list_limit = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_POISON")) ctx->poison = (int)std::strtol(env, nullptr, 10);
    if (const char* env = std::getenv("RTMI_TRACE_DIAG")) ctx->trace_diagnostics = env[0] == '1';
    if (const char* env = std::getenv("RTMI_CALIBRATION_LOG")) ctx->calibration_log = env[0] == '1';
    if (const char* env = std::getenv("RTMI_RECALIBRATE")) ctx->recalibrate = env[0] == '1';
    if (const char* env = std::getenv("RTMI_REFIT")) ctx->refit_choice = env[0] == 'r' ? 1 : env[0] == 'c' ? 2 : 0;
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
    return RT_SUCCESS;
}

int rt_buf_upload(rt_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !h_src))) return RT_FAIL(ctx, "null argument");
    if (!bytes) return RT_SUCCESS;
    RT_ENTER(ctx);
    RT_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // the host buffer may be reused on return
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
    if (parity_mode != RT_MODE_FIXED && parity_mode != RT_MODE_LITERAL_ZERO) return RT_FAIL(ctx, "unknown parity mode");
    rt_bvh* b = new (std::nothrow) rt_bvh();
    if (!b) return RT_FAIL(ctx, "out of host memory");
    b->ctx = ctx; b->scene = scene; b->mode = parity_mode;
    *out = b;
    return RT_SUCCESS;
}

static void block_order_free(BlockOrder& o);

static void free_build_buffers(rt_bvh* b)
{
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
    block_order_free(b->block_order);
    delete b;
}

// Large scenes refit by range queries instead of climbing parent links (build_kernels.hip 4a): the climb is serial in the depth of
// the reference tree, which duplicate Morton codes turn into chains of thousands of levels.  Same node bytes either way; below
// the threshold the single climbing launch is faster than the dozen small launches of the query structure.
// env RTMI_REFIT=rq / climb forces one (tests run both on every fixture).
static bool refit_by_range_queries(const rt_ctx* ctx, uint32_t n, int mode)
{
    if (mode != RT_MODE_FIXED || n < 256u) return false;
    if (ctx->refit_choice) return ctx->refit_choice == 1;
    return n >= (1u << 17);
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
        const bool want_search = !ctx->no_search_tree && b->mode == RT_MODE_FIXED;
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
            b->quad_cam_variants = (one * 9 <= ((size_t)4 << 30) && !std::getenv("RTMI_NO_OCTANT_RECORDS")) ? 9u : 1u;
            e = hipMalloc((void**)&b->d_quad_cam, one * b->quad_cam_variants);
        }
        if (e == hipSuccess && !ctx->no_packet4) e = hipMalloc((void**)&b->d_tricam, (size_t)n * sizeof(TriCam));
        if (e == hipSuccess && want_search) {   // working set of the search-tree build, kept between builds (an allocation costs ~0.1 ms)
            e = hipMalloc((void**)&b->s_nodes, (size_t)(2 * n - 1) * sizeof(rt_bvh_node));
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_pairs, (size_t)N * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_counters, (size_t)n * sizeof(uint32_t) + 256);
            if (e == hipSuccess) e = hipMalloc((void**)&b->s_pos, (size_t)n * sizeof(uint32_t));
        }
        if (e == hipSuccess && refit_by_range_queries(ctx, n, b->mode)) {
            e = hipMalloc((void**)&b->d_spans, (size_t)(n - 1) * sizeof(uint2));
            if (e == hipSuccess) e = hipMalloc(&b->d_rq_scratch, refit_rq_scratch_bytes(n));
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
    RT_HIP(ctx, hipMemsetAsync(b->d_pairs, 0xFF, (size_t)N * sizeof(uint2), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(b->d_counters, 0, (size_t)n * sizeof(uint32_t), ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(b->d_nodes, 0, (size_t)(2 * n - 1) * sizeof(rt_bvh_node), ctx->stream));
    b->frame_initialised = true;
    b->built = false;
    return RT_SUCCESS;
}

int rt_bvh_construct(rt_bvh* b)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->frame_initialised) return RT_FAIL(ctx, "initializeFrame must precede construct");
    RT_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const DeviceScene& sc = b->scene->d;
    launch_morton_leaves(s, sc, b->d_nodes, b->d_pairs);          // 1. BVHManager.cpp:183
    launch_bitonic_sort(s, b->d_pairs, b->N);                     // 2. :195
    const bool literal = b->mode == RT_MODE_LITERAL_ZERO;
    launch_radix_tree(s, b->d_nodes, b->d_pairs, b->n, literal ? b->n - 2u : b->n - 1u, b->d_spans);   // 3. :204 (LITERAL_ZERO: the guard as written, BVHKernels.cl:58)
    if (literal) launch_refit_literal(s, b->d_nodes, b->d_counters, b->n, b->d_depth);
    else if (b->d_spans) {
        // 4. by range queries; NaN boxes (not associative) send it back to the climb
        launch_refit_rq(s, b->d_nodes, b->d_pairs, b->d_spans, b->n, b->d_rq_scratch, b->d_depth, b->d_depth + 1);
        uint32_t nan_boxes = 0;
        RT_HIP(ctx, hipMemcpyAsync(&nan_boxes, b->d_depth + 1, 4, hipMemcpyDeviceToHost, s));
        RT_HIP(ctx, hipStreamSynchronize(s));
        if (nan_boxes) launch_refit(s, b->d_nodes, b->d_counters, b->n, b->d_depth);
    }
    else launch_refit(s, b->d_nodes, b->d_counters, b->n, b->d_depth); // 4. :217 (+ the tree depth)
    launch_relayout(s, b->d_nodes, b->d_pairs, b->d_trav, b->d_pos, b->n);  // traversal records, DFS ranks
    launch_pack_triangles(s, sc, b->d_pos, b->d_trirec);
    RT_HIP(ctx, hipGetLastError());
    uint32_t depth = 0;
    RT_HIP(ctx, hipMemcpyAsync(&depth, b->d_depth, 4, hipMemcpyDeviceToHost, s));
    // search tree for primary rays: same leaves, better order
    uint32_t search_depth = 0;
    b->have_search_tree = false;
    const bool want_search = !ctx->no_search_tree && !literal;
    if (want_search) {
        const uint32_t n = b->n, N = b->N;
        rt_bvh_node* s_nodes = b->s_nodes; uint2* s_pairs = b->s_pairs; uint32_t* s_counters = b->s_counters; uint32_t* s_pos = b->s_pos;
        RT_HIP(ctx, hipMemsetAsync(s_nodes, 0, (size_t)(2 * n - 1) * sizeof(rt_bvh_node), s));
        RT_HIP(ctx, hipMemcpyAsync(s_nodes, b->d_nodes, (size_t)n * sizeof(rt_bvh_node), hipMemcpyDeviceToDevice, s));   // the leaves
        RT_HIP(ctx, hipMemsetAsync(s_pairs, 0xFF, (size_t)N * sizeof(uint2), s));
        RT_HIP(ctx, hipMemsetAsync(s_counters, 0, (size_t)n * sizeof(uint32_t) + 256, s));
        launch_search_tree_keys(s, b->d_nodes, n, s_pairs);
        launch_bitonic_sort(s, s_pairs, N);
        launch_radix_tree(s, s_nodes, s_pairs, n, n - 1u);
        uint32_t* s_depth = s_counters + n;                     // spare words behind the counters
        launch_refit(s, s_nodes, s_counters, n, s_depth, !ctx->search_tree_reference_boxes);
        launch_relayout(s, s_nodes, s_pairs, b->d_trav + (n - 1), s_pos, n, n - 1);
        RT_HIP(ctx, hipGetLastError());
        RT_HIP(ctx, hipMemcpyAsync(&search_depth, s_depth, 4, hipMemcpyDeviceToHost, s));
    }
    if (b->d_quad) launch_relayout_quad(s, b->d_trav, b->d_quad, (b->n - 1) * (want_search ? 2u : 1u));
    RT_HIP(ctx, hipStreamSynchronize(s));                          // construct() returns with the build finished
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
    return RT_SUCCESS;
}

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
static uint32_t strategy_parts(const rt_bvh* b, const rt_camera* cam, bool packet, bool quad)
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
#define RT_BLOCK_ORDER_REFRESH 32u          // frames between two sorts (one launch of one workgroup, k_block_order)
static void block_order_free(BlockOrder& o)
{
    if (o.d_cost) (void)hipFree(o.d_cost);
    if (o.d_order) (void)hipFree(o.d_order);
    o = BlockOrder();
}
// buffers for a frame of `blocks` blocks; false: none (allocation failed): the frame runs in index order
static bool block_order_fit(rt_ctx* ctx, BlockOrder& o, uint32_t blocks)
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
static void block_order_refresh(rt_ctx* ctx, BlockOrder& o)
{
    o.valid = launch_block_order(ctx->stream, o.d_cost, o.d_order, o.count, ctx->block_order_group_log, false);
    o.age = 0;
}

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
};

static TraceLaunch make_launch(rt_bvh* b, const rt_camera* cam = nullptr, uint32_t parts = 0, const FrameMode* mode = nullptr)
{
    TraceLaunch t{};
    t.trav = b->d_trav; t.search_root = b->have_search_tree ? b->n - 1 : 0u; t.tree_select = b->d_tree_select; t.scene = &b->scene->d; t.pos = b->d_pos; t.trirec = b->d_trirec; t.tricam = b->d_tricam; t.n_leaves = b->n;
    t.ctrl = b->ctx->d_trace_ctrl; t.long_list = b->ctx->d_long_list; t.long_cap = b->ctx->long_cap;
    t.defer_budget = b->ctx->defer_budget;
    t.packet = false; t.packet4 = false; t.tag = b->ctx->profiling_tag; t.diagnostics = b->ctx->trace_diagnostics;
    t.quad = b->d_quad; t.quad_cam = b->d_quad_cam; t.quad_cam_variants = b->quad_cam_variants; t.no_unsafe_rays = b->capture_no_unsafe_rays;
    t.waves_per_block = (mode && mode->fixed_strategy) ? mode->waves_per_block
                        : b->ctx->trace_workgroup_waves ? b->ctx->trace_workgroup_waves : (b->ctx->frames_in_flight > 1u ? 1u : 0u);
    t.keys = b->ctx->d_keys; t.split_list = b->ctx->d_split_list; t.pairs = b->d_pairs;
    t.task_shards = b->ctx->d_task_list; t.task_cap = b->ctx->task_cap; t.task_helper_blocks = b->ctx->task_helper_blocks;
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
static int ensure_host_planes(rt_bvh* b);
static uint32_t host_plane_bits(const rt_bvh* b, const float origin[3]);
// eager four-wide frames on one build before the host reads the leaf boxes' planes back (once per build: a synchronisation, three
// sorts on the device and their copies -- 11 ms at 70 000 leaves, which a scene that is rebuilt every frame or lives for a few
// frames never pays; it pays back 2-6 us per frame and the marks of a moved camera)
#ifndef RT_EAGER_PLANES_AFTER
#define RT_EAGER_PLANES_AFTER 32u
#endif

// the table work_to_pixel_block reads: (x0, y0) of tile `ordinal` of this rank for (resolution, rank, tiling); null when the context
// already holds RT_TILE_TABLES shapes or the allocation fails (the kernels then compute it per lane)
static const uint2* tile_table(rt_ctx* ctx, const rt_camera* cam, uint32_t rank, uint32_t nranks)
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

static int trace_primary_tuned(rt_bvh* b, const rt_camera* cam, uint32_t rank, uint32_t nranks, bool compact, rt_contact* out, const FrameMode* mode = nullptr)
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
        // Deadlines of the candidates on the device's own clock (trace_kernels.hip, cal_expired): a sample of that clock next to a
        // host time stamp maps "now + limit" onto it.  The sample is taken behind the warm-up launch; its kernel ran before the
        // read-back returned, so the mapped deadlines come a few tens of microseconds early: a slack is added.
        unsigned long long clock0 = 0; double clock_rate = 0.0;        // ticks per millisecond
        std::chrono::steady_clock::time_point host0;
        float limit_ms = 0.f;                                          // 0 = the next launch has no deadline
        auto sample_clock = [&]() {
            static const bool off = [] { const char* e = std::getenv("RTMI_CALIBRATION_DEADLINE"); return e && e[0] == '0'; }();
            if (off) return;
            if (!ctx->d_clock && hipMalloc((void**)&ctx->d_clock, 8) != hipSuccess) { (void)hipGetLastError(); ctx->d_clock = nullptr; return; }
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); return; }
            launch_device_clock(ctx->stream, ctx->d_clock);
            unsigned long long v = 0;
            if (hipMemcpyAsync(&v, ctx->d_clock, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return; }
            host0 = std::chrono::steady_clock::now();
            clock0 = v; clock_rate = (double)khz;
        };
        // one launch of one candidate; false = runtime failure
        auto run_once = [&](int ci, float& ms, bool& valid) -> bool {
            const auto& cand = candidates[ci];
            TraceLaunch t = make_launch(b, cam, cal_parts);              // (the camera is prepared by the first of them)
            t.packet = cand.packet; t.packet4 = cand.quad;
            t.defer_budget = cand.budget;
            t.tag = 1;
            if (cand.packet && cand.quad && order_ok) { t.block_cost = bo->d_cost; t.block_order = bo->valid ? bo->d_order : nullptr; }
            if (clock_rate > 0.0 && limit_ms > 0.f) {
                const double now_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host0).count();
                t.cal_deadline = clock0 + (unsigned long long)((now_ms + (double)limit_ms + 0.1) * clock_rate);
            }
            ms = 0.f;
            (void)hipEventRecord(e0, ctx->stream);
            t.tile_xy = tile_table(ctx, cam, rank, nranks);
            launch_trace_primary(ctx->stream, t, *cam, rank, nranks, compact, out);
            (void)hipEventRecord(e1, ctx->stream);
            launches++;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
            gpu_ms += ms;
            uint32_t cand_err = 0;
            if (hipMemcpyAsync(&cand_err, (const char*)ctx->d_trace_ctrl + trace_ctrl_error_offset(), 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
            valid = cand_err == 0;                                       // e.g. a list too small for this strategy: not a candidate
            if (cand_err && (hipMemsetAsync((char*)ctx->d_trace_ctrl + trace_ctrl_error_offset(), 0, 4, ctx->stream) != hipSuccess ||
                             hipStreamSynchronize(ctx->stream) != hipSuccess)) return false;   // the word is sticky: a discarded candidate must not fail the frame
            last_failed = !valid;
            if (ctx->calibration_log)
                std::fprintf(stderr, "rtmi calibration: n=%u %ux%u %s budget %u: %.4f ms%s\n", b->n, cam->resX, cam->resY,
                             cand.packet ? (cand.quad ? "packet4" : "packet") : "lane-per-ray", cand.budget, ms, valid ? "" : " (error word set, discarded)");
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
            static const float in_flight = [] { const char* e = std::getenv("RTMI_IN_FLIGHT_SHARE_HANDICAP"); const float v = e ? (float)std::atof(e) : 0.f; return v >= 1.0f ? v : 1.6f; }();
            if (ctx->frames_in_flight > 1u) h *= in_flight;
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
            float m = 0.f; bool valid = false;
            if (!run_once(ci, m, valid)) { broken = true; return; }
            if (!valid) { cand_n[ci] = -1; return; }
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
        static const bool calibrate_all = [] { const char* e = std::getenv("RTMI_CALIBRATE_ALL"); return e && e[0] == '1'; }();
        // The first candidate runs without a deadline, twice (an untimed warm-up, then its sample), so it should never be a
        // hopeless one: sharing after 64 steps has no tail (since the search tree's boxes are tight a 1080p block takes 3-13 steps on
        // average and 50-900 at most: Tenryuu whole blocks 0.173 ms, shared after 256 / 128 / 64 / 32 steps 0.195 / 0.18 / 0.132 /
        // 0.14; ManFigure 0.09 / 0.108 from 64 up; ClumsyDragon 0.61 / 0.40 / 0.31 / 0.25 / 0.22 against 0.127 for lane-per-ray 16),
        // whole blocks and the lane-per-ray budgets can be several times behind.
        const int first_cand = cand_n[P4_64] == 0 ? P4_64 : cand_n[P4_NONE] == 0 ? P4_NONE : cand_n[LANE_128] == 0 ? LANE_128 : -1;
        {
            int warm = first_cand;
            for (int ci = 0; ci < N_CAND && warm < 0; ci++) if (cand_n[ci] == 0) warm = ci;
            if (warm >= 0) { float m; bool v = false; if (!run_once(warm, m, v)) broken = true; }
            // the warm-up ran to its end: its block costs give the candidates the start order the frames will have
            if (!broken && warm >= 0 && candidates[warm].packet && candidates[warm].quad && order_ok) block_order_refresh(ctx, *bo);
            if (!broken) sample_clock();
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
        const int best_ci = broken ? -1 : best_of(-1);
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

int rt_bvh_generate_contacts_rays(rt_bvh* b, const void* d_rays, uint32_t ray_count, void* d_contacts)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (ray_count == 0) return RT_SUCCESS;
    if (!d_rays || !d_contacts) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, ray_count)) return RT_ERROR;
    launch_trace_rays(ctx->stream, make_launch(b), (const rt_ray*)d_rays, ray_count, (rt_contact*)d_contacts);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

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

int rt_bvh_info(const rt_bvh* b, uint32_t* leaves, uint32_t* morton_items, uint32_t* max_depth)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    if (leaves) *leaves = b->n;
    if (morton_items) *morton_items = b->N;
    if (max_depth) *max_depth = b->max_depth;
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
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void current_strategy(const rt_bvh* b, bool& packet, bool& quad)
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
static int ensure_host_planes(rt_bvh* b)
{
    rt_ctx* ctx = b->ctx;
    if (b->h_planes_generation == b->build_generation && !b->h_planes[0].empty()) return RT_SUCCESS;
    // sorted on the device (launch_plane_keys), copied back per axis; FLT_MIN joins them here (BVH.h:310)
    static const bool dbg_t = [] { const char* e = std::getenv("RTMI_GRAPH_DEBUG"); return e && e[0] == '1'; }();
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
static uint32_t host_plane_bits(const rt_bvh* b, const float origin[3])
{
    uint32_t bits = 0;
    for (int k = 0; k < 3; k++)
        if (origin[k] == origin[k] && std::binary_search(b->h_planes[k].begin(), b->h_planes[k].end(), origin[k])) bits |= 1u << k;
    return bits;
}

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
    m->fused = (b->capture_no_unsafe_rays || (m->fixed_strategy && !m->packet)) && !ctx->no_fused_frames ? &fused : nullptr;
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
        static const bool want_direct = [] { const char* e = std::getenv("RTMI_GRAPH_DIRECT"); return !(e && e[0] == '0'); }();
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
    // per-octant records: on a y or z plane): the eager frame above has computed those bits for this camera position, so a
    // captured frame of a camera on no such plane leaves the (then empty, ~7 us) wave-per-ray launch out.  When the camera
    // moves (rt_frame_graph_set_camera) the bits are looked up on the host.
    uint32_t plane_bits = 7u;
    if (b->d_tree_select && b->have_search_tree) {
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
    static const bool debug = [] { const char* e = std::getenv("RTMI_GRAPH_DEBUG"); return e && e[0] == '1'; }();
    if (debug && std::getenv("RTMI_GRAPH_DEBUG_VERBOSE"))
        std::fprintf(stderr, "rtmi graph %p: launch exec[%d] dirty %d parts %#x marked %#x same %d strategy %d%d/%u no_unsafe %d plane %#x origin %.9g %.9g %.9g\n", (void*)g, which,
                     (int)g->dirty[which][g->slot[which]], g->parts, b->marked_parts, (int)same, (int)g->packet, (int)g->quad, g->budget, (int)g->no_unsafe, g->plane_bits_known,
                     g->origin[0], g->origin[1], g->origin[2]);
    // the instance to launch: the one launched last if its arguments are current, else the other one (its last launch is the older
    // one); its arguments are re-set only behind that launch
    int k = g->slot[which];
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
    g->has_event[which][k] = g->moved;
    if (g->moved) RT_HIP(ctx, hipEventRecord(g->done[which][k], ctx->stream));
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
    if (g->host_launches && std::getenv("RTMI_GRAPH_DEBUG"))
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
