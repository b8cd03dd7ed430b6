// rtmi_internal.h -- shared declarations of librtmi.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

// ---------------------------------------------------------------------------------------
// Device-side data layout (all in HBM, built once per construct(), read-only while tracing)
// ---------------------------------------------------------------------------------------

// One 64-byte record per INNER node i (reference node index n+i): both child boxes plus the
// child ids, so one visit = one 64-byte fetch instead of the reference's 48 + 32 + 32 bytes
// (BVH.h:359-364).  Child id: bit31 set -> leaf (low bits = leaf/global triangle index),
// else inner record index.  The boxes are bit copies of the reference BVHNode boxes.
struct __attribute__((aligned(16))) TravNode {
    float a_lo[3]; uint32_t child_a;
    float a_hi[3]; uint32_t child_b;
    float b_lo[3]; uint32_t pad0;
    float b_hi[3]; uint32_t pad1;
};
static_assert(sizeof(TravNode) == 64, "TravNode must be 64 bytes");

// Four-wide view of the same hierarchies for the packet kernel: record i = inner node i with its inner children replaced
// by THEIR children (2-4 slots).  q[0..7] as scalar-loadable float4: (lo0,id0) (hi0,id1) (lo1,id2) (hi1,id3) (lo2,marks)
// (hi2,-) (lo3,-) (hi3,-); ids are the binary record indices / RT_LEAF_FLAG|leaf, empty slot = RT_SENTINEL.
struct __attribute__((aligned(16))) QuadNode { float4 q[8]; };
static_assert(sizeof(QuadNode) == 128, "QuadNode must be 128 bytes");

// One 48-byte record per triangle for the packet kernels, which fetch a leaf's data through the scalar path: everything the
// triangle test needs behind ONE address (vertex 0, both edges) plus the leaf's depth-first rank (the tie rule) -- instead of three
// arrays (three address computations, six scalar loads) and a vector load of the rank in the tie case.
struct __attribute__((aligned(16))) TriRec { float v0[3]; float e1[3]; float e2[3]; uint32_t pos; uint32_t material; uint32_t pad; };
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");
// The triangle test of PRIMARY rays, which share their origin (the camera): what Triangle.h:66-107 computes from the origin and
// the triangle alone -- tvec = origin - vertex 0, qvec = tvec x edge 1, edge 2 . qvec -- is the same for every ray of a frame
// and is computed once per camera position (k_prepare_camera), with the same operations in the same order.
struct __attribute__((aligned(16))) TriCam { float e1[3]; float e2[3]; float tvec[3]; float qvec[3]; float e2_dot_qvec; uint32_t pos; uint32_t pad[2]; };
static_assert(sizeof(TriCam) == 64, "TriCam must be 64 bytes");

#define RT_LEAF_FLAG 0x80000000u
#define RT_SENTINEL 0xFFFFFFFFu   // stack sentinel, BVH.h:352
#define RT_LONG_LIST_MAX (1u << 20)     // at most this many deferred-ray records (128 MiB)

// One entry per mesh of the packed scene buffer (host walks the headers once at upload).
struct MeshEntry {
    uint64_t byte_offset;     // of the MeshHeader inside the scene buffer
    uint32_t first_triangle;  // global index of the mesh's first triangle
    uint32_t triangle_count;
    uint32_t vertex_count;
    uint32_t material_index;  // (CL_UINT)MeshHeader.materialIndex, BVH.h:395
    uint32_t model_index;
    uint32_t submesh_index;
};

struct DeviceScene {
    const unsigned char* bytes;     // raw packed scene buffer (reference format)
    uint64_t nbytes;
    uint32_t n_tris;
    uint32_t n_meshes;
    const MeshEntry* meshes;
    // SoA triangle table indexed by global triangle index == BVH leaf index
    float4* v0;        // xyz = vertex 0, w = material index (bits)
    float4* v1;
    float4* v2;
    float4* e1;        // v1 - v0 (Triangle.h:72)
    float4* e2;        // v2 - v0 (Triangle.h:73)
    float4* nrm;       // normalize(cross(e1,e2)) (Triangle.h:105), ray independent
    uint4*  ref;       // (triangle in mesh, submesh, model, material)
    // scene header fields needed on the device
    uint32_t index_bytes;      // width of a vertex index in the packed buffer: 2 (reference) or 4 (rt_scene_upload_indexed)
    uint32_t n_lights, n_spheres, n_materials;
    float bb_lo[3], bb_hi[3];
};

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipDeviceProp_t prop;
    mutable std::string last_error;
    // shared state of the trace kernels (one launch at a time per context)
    void* d_trace_ctrl = nullptr;
    uint32_t* d_long_list = nullptr;      // deferred-ray records, grow-only
    uint32_t long_cap = 0;
    unsigned long long* d_keys = nullptr;   // packet kernel: (t, position) keys of split rays, one per work item, grow-only
    uint64_t keys_cap = 0;
    uint32_t* d_split_list = nullptr;       // long_cap entries
    void* d_task_list = nullptr;            // trace_task_bytes(task_cap): sub-tree tasks shared between the waves of k_trace_packet4
    uint32_t task_cap = 0, task_helper_blocks = 0;
    bool task_cap_auto = false;
    uint32_t share_min_budget = 0;          // calibration: sharing candidates with a smaller budget are not tried
    bool no_fused_frames = false;           // knob RTMI_FUSED_FRAMES=0: frame graphs always shade with k_shade
    // the knobs below used to be read where they are used, once per process; they are read with the others in rt_ctx_create (rtmi_capi.hip:
    // the ONE place of the shipped library that looks at the environment), per context
    bool quad_by_area = true;               // RTMI_QUAD_BY_AREA=0: four-wide records open both children instead of the largest slot twice
    bool no_octant_records = false;         // RTMI_NO_OCTANT_RECORDS: one camera-relative copy of the four-wide records instead of nine
    bool no_build_graph = false;            // RTMI_BUILD_GRAPH=0: every construct() is enqueued launch by launch
    bool no_graph_direct = false;           // RTMI_GRAPH_DIRECT=0: frame graphs of kernels only go through hipGraphLaunch as well
    int graph_debug = 0;                    // RTMI_GRAPH_DEBUG=1 (+ RTMI_GRAPH_DEBUG_VERBOSE: 2): host time per stage of the frame graphs, on stderr
    std::string rays_forced;                // RTMI_RAYS=lane:N | p4 | p4s:N: the ray-buffer entry's strategy, forced
    // what the library's own generators last wrote to a ray buffer (rtmi_rays.hip: part of the ray-buffer entry's calibration key)
    struct RayTag { const void* addr = nullptr; uint8_t kind = 0; };
    RayTag ray_tags[8];
    uint32_t ray_tags_next = 0;
    bool tlg_no_rebuild_on_previous = false;// RTMI_TLG_REBUILD_ON_PREVIOUS_SIZES=0
    bool tlg_no_tricam = false;             // RTMI_TLG_TRICAM=0
    bool tlg_no_walk_records = false;       // RTMI_TLG_WALK_RECORDS=0: the grid walk of round 4 (no cell / plane records, IEEE divisions in every set-up)
    bool no_calibration_deadline = false;   // RTMI_CALIBRATION_DEADLINE=0
    float in_flight_share_handicap = 1.6f;  // RTMI_IN_FLIGHT_SHARE_HANDICAP
    bool calibrate_all = false;             // RTMI_CALIBRATE_ALL=1
    bool search_keys_ten_per_axis = false;  // RTMI_SEARCH_KEYS=cubic: the search tree's keys with ten bits per axis whatever the box's shape (rounds 1-4)
    uint32_t long_batch = 0;                // RTMI_LONG_BATCH=1|2|4: rays per wave of the wave-per-ray kernel (0: by the number of deferred rays)
    uint64_t scratch_generation = 0;        // bumped whenever one of the three buffers above is reallocated (frame graphs hold their addresses)
    bool trace_launched = false;  // a trace launch has not been checked for its error word yet
    uint32_t defer_budget = 0;    // tuning knob (env RTMI_DEFER_BUDGET), 0 = built-in default
    bool profiling = false;       // rt_ctx_set_profiling: HIP events around the two trace kernels
    int  profiling_tag = 0;       // 2: launches use the <.., 2> kernel instances (isolated measurement pass)
    bool force_packet = false, no_packet = false;   // tuning knobs (env RTMI_PACKET=1 / 0; 4 = the four-wide packet kernel)
    bool force_packet4 = false, no_packet4 = false; // env RTMI_PACKET=4 / RTMI_PACKET4=0 (no four-wide records at all)
    int poison = 0;               // test knob (env RTMI_POISON=bits): new deferred-list (1) / split-list (2) / key (4) buffers are filled with 0xCD
    bool no_search_tree = false;  // knob (env RTMI_SEARCH_TREE=0): primary rays walk the reference tree
    bool trace_diagnostics = false; // knob (env RTMI_TRACE_DIAG=1)
    int refit_choice = 0;         // knob (env RTMI_REFIT=rq / climb / span): 1 range-query refit (round 2's sparse table), 2 climbing refit, 3 / 0 refit by spans (from 256 triangles on)
    bool recalibrate = false;     // knob (env RTMI_RECALIBRATE=1): every construct() forgets the calibrated trace strategy
    bool calibration_log = false; // knob (env RTMI_CALIBRATION_LOG=1): one stderr line per calibration candidate
    uint32_t long_list_limit = 0; // test knob (env RTMI_LONG_LIST_MAX): caps the deferred-ray list, 0 = RT_LONG_LIST_MAX
    std::vector<hipEvent_t> trace_events;   // triples, one per trace launch since the last query
    unsigned long long* d_clock = nullptr;  // one word: the device clock sample of a calibration
    // where a rank's owned tiles lie, per frame shape (trace_kernels.hip, work_to_pixel_block): small device tables kept for the context's
    // life (captured kernels hold their addresses); beyond RT_TILE_TABLES shapes the kernels compute the place per lane as before
    struct TileTable { uint32_t resX = 0, resY = 0, rank = 0, nranks = 0; uint2* d = nullptr; };
    enum { RT_TILE_TABLES = 16 };
    TileTable tile_tables[RT_TILE_TABLES];
    uint32_t n_tile_tables = 0;
    bool search_tree_reference_boxes = false;   // knob RTMI_SEARCH_TREE_BOXES=reference: the search tree's boxes clamped like the reference's (round 2's)
    bool no_block_order = false;            // knob RTMI_BLOCK_ORDER=0: waves walk the blocks in index order
    uint32_t trace_workgroup_waves = 0;     // knob RTMI_TRACE_WORKGROUP=64/128/256: waves per workgroup of the first-phase kernels (0: by policy)
    bool order_in_flight = true;            // knob RTMI_BLOCK_ORDER_IN_FLIGHT=0: index order under the throughput policy (what paid on round 2's tree)
    bool older_runtime = false;             // the HIP runtime serving this process is older than the one librtmi.so was built against (torch's bundled copy loaded first)
    uint32_t frames_in_flight = 1;          // rt_ctx_set_frames_in_flight: > 1 = throughput policy (index order, sharing handicapped)
    uint32_t block_order_group_log = 4;     // knob RTMI_BLOCK_ORDER_GROUP=1..64: consecutive blocks that stay together (a power of two)
};

// Start order of a frame's 8x8 blocks by their cost in an earlier frame (trace_kernels.hip, launch_block_order): the buffers of one
// frame shape -- the BVH's for eager frames, a frame graph's own for captured ones (their addresses are kernel arguments)
struct BlockOrder {
    uint32_t* d_cost = nullptr; uint32_t* d_order = nullptr;
    uint32_t cap = 0;                       // blocks the buffers hold
    uint32_t count = 0;                     // blocks of the frame shape the order is for
    bool valid = false;                     // d_order holds a permutation of [0, count)
    uint32_t age = 0;                       // frames since the order was last derived from the costs
};

struct rt_scene {
    rt_ctx* ctx = nullptr;
    DeviceScene d{};
    std::vector<MeshEntry> h_meshes;
    void* d_bytes = nullptr;
    void* d_meshes = nullptr;
    void* d_tri = nullptr;     // one allocation backing the SoA arrays
    float bb_w[2] = { 0, 0 };  // .w of SceneHeader.modelsBoundingBox bounds (carried into GridData.box verbatim)
    bool has_nan = false;      // some vertex coordinate is NaN (seen at upload): only then can a leaf box be NaN and a refit by spans need its fall-back
};

struct rt_bvh {
    rt_ctx* ctx = nullptr;
    const rt_scene* scene = nullptr;
    int mode = 0;              // RT_MODE_FIXED / LITERAL_ZERO / FAST
    bool device_arith = false; // ... | RT_MODE_DEVICE_ARITH
    uint32_t n = 0;            // leaves
    uint32_t N = 0;            // morton buffer items (power of two)
    rt_bvh_node* d_nodes = nullptr;    // reference layout, 2n-1 entries
    uint2* d_pairs = nullptr;          // N (morton, leaf)
    uint32_t* d_counters = nullptr;    // n visit counters (BVHManager.cpp:67)
    TravNode* d_trav = nullptr;        // traversal records: reference tree, then the search tree
    QuadNode* d_quad = nullptr;        // four-wide records of the same nodes (packet kernel), null when disabled
    QuadNode* d_quad_cam = nullptr;    // the same records relative to the current camera position (k_prepare_camera): the general copy, then
    uint32_t quad_cam_variants = 1;    // (quad_cam_variants - 1 = 8) copies with the near / far planes in place per direction octant
    bool capture_no_unsafe_rays = false;   // set around a frame-graph capture whose camera is known to sit on no box plane
    uint32_t* d_depth = nullptr;       // [0] max depth, [1] NaN flag of the span refit; [2], [3]: the same of the search tree
    uint32_t* d_pos = nullptr;         // leaf -> position in d_pairs (DFS rank of the leaf)
    TriRec* d_trirec = nullptr;        // packed triangle records of the packet kernels (filled per build: they hold the rank)
    TriCam* d_tricam = nullptr;        // ... relative to the current camera position (with the four-wide records): k_prepare_camera
    bool have_search_tree = false;     // d_trav holds 2(n-1) records: reference tree, then the search tree (primary rays)
    uint32_t* d_tree_select = nullptr; // 1 word, bit k: the camera's coordinate k equals a box plane (k_prepare_camera)
    uint32_t search_depth = 0;
    rt_bvh_node* s_nodes = nullptr;    // working set of the search-tree build (kept between builds)
    uint2* s_pairs = nullptr;
    uint32_t* s_counters = nullptr;
    uint32_t* s_pos = nullptr;
    // construct() as a HIP graph (rtmi_bvh.hip): the sequence of one build on this BVH's buffers, captured on the second build of an
    // allocation and replayed by the later ones (a scene rebuilt every frame, OpenCLRayTracerDemo.cpp:230-234)
    hipGraph_t build_graph = nullptr; hipGraphExec_t build_exec = nullptr;
    bool build_graph_failed = false;        // this allocation's build could not be captured / instantiated: built eagerly from then on
    uint2* d_sort_tmp2 = nullptr; uint32_t* d_sort_scratch2 = nullptr;      // RT_MODE_FAST: the reference tree's radix sort has its own scratch (both sorts may run at once)
    uint32_t builds_on_allocation = 0;      // construct() calls since the buffers were (re)allocated
    uint32_t* h_depths = nullptr;           // pinned copy of d_depth[0..3] (read back at the end of a build)
    bool clears_pending = false;            // initializeFrame's clears, enqueued with the build
    uint32_t graph_builds = 0;              // construct() calls served by the graph (inspection)
    uint2* d_sort_tmp = nullptr;       // radix sort: n pairs to ping-pong with, then its histograms (launch_radix_sort)
    uint32_t* d_sort_scratch = nullptr;
    uint2* s_spans = nullptr;          // the search tree's spans + scratch of its refit by spans
    void* s_span_scratch = nullptr;
    void* d_span_scratch = nullptr;    // the reference tree's (refit by spans)
    uint2* d_spans = nullptr;          // range-query refit: [first, last] of every inner node + its scratch (large scenes only)
    void* d_rq_scratch = nullptr;
    uint32_t* d_packet_spill = nullptr;   // packet kernels: bitmap + stack overflow areas, sized from the tree depth
    uint32_t packet_spill_entries = 0;
    uint32_t packet_spill_slots = 0;      // overflow areas = resident waves the device can hold (multiple of 32)
    uint32_t max_depth = 0;
    float marked_origin[3] = { 0, 0, 0 };   // camera position the per-camera records were computed for
    bool marked_valid = false;
    uint32_t marked_parts = 0;              // ... and which of them (RT_PREP_* bits): a strategy prepares only what it reads
    // planes of the leaf boxes per axis, sorted, on the host (+ FLT_MIN, BVH.h:310): "does this camera coordinate equal a box
    // plane" without asking the device -- read back on first use per build (movable frame graphs, rt_frame_graph_set_camera)
    std::vector<float> h_planes[3];
    uint64_t h_planes_generation = 0;
    uint32_t h_planes_reads = 0;            // how often the leaves were read back for them (rt_bvh_host_plane_reads: tests)
    bool tuned = false;                     // deferral budget calibrated for tuned_key
    uint64_t tuned_key = 0;
    uint32_t tuned_n = 0;                   // triangle count the strategy was calibrated for (kept across rebuilds of the same size)
    uint32_t tuned_budget = 0;              // 0 = packet kernel, else deferral budget of the two-phase kernels
    bool tuned_packet = false;
    bool tuned_quad = false;                // with tuned_packet: the four-wide packet kernel
    // the last few other calibrations (resolution / tiling keys): a BVH that serves several viewports or a tiled and an untiled frame
    // in turns does not calibrate again at every switch
    struct TunedEntry { uint64_t key = 0; uint32_t budget = 0; bool packet = false, quad = false, valid = false; };
    TunedEntry tuned_cache[4];
    uint32_t tuned_cache_next = 0;
    BlockOrder block_order;                 // eager frames
    uint32_t calibration_launches = 0;      // full-size launches the last calibration took (diagnostic)
    float calibration_gpu_ms = 0.f;         // ... and their device time
    // the ray-buffer entry (rtmi_rays.hip): kernel and budget calibrated on the first large buffer traced through this BVH
    // ... one entry per size class (power of two) of the buffers: the buffers of one bounce of a frame loop keep their size class and their
    // character from frame to frame (a camera's 2 M rays, the 0.8 M reflections of their hits, ...), different bounces seldom share one
    // ... keyed further by what the buffer holds, as far as the library knows it (a strategy calibrated on a camera's rays served the
    // reflections of the same size class): buffers written by rt_generate_primary_rays / rt_generate_secondary_rays carry their kind
    // (rt_ctx::ray_tags), a caller's own buffer is known by its address; and an entry whose launches come to take twice its calibrated time
    // (events around every tuned launch, read without waiting by the next call) is calibrated again
    struct RaysTuned { uint32_t size_class = 0, budget = 0; bool packet = false, valid = false; uint8_t kind = 0; const void* addr = nullptr;
                       float ms = 0.f; uint32_t slow = 0; bool pending = false; hipEvent_t e0 = nullptr, e1 = nullptr; };
    enum { RT_RAYS_TUNED = 8 };
    RaysTuned rays_tuned[RT_RAYS_TUNED];
    uint32_t rays_tuned_next = 0, rays_tuned_n = 0;
    int rays_last = -1;                      // entry the last large buffer was traced with (rt_bvh_rays_strategy)
    uint32_t rays_calibration_launches = 0; float rays_calibration_gpu_ms = 0.f;     // of the last calibration
    uint32_t rays_recalibrations = 0;        // entries retired because their launches came to take twice their calibrated time
    uint64_t build_generation = 0;          // bumped by every construct() and every reallocation (frame graphs hold record addresses)
    uint32_t eager_frames = 0;              // eager four-wide frames traced on this build (from RT_EAGER_PLANES_AFTER on the host looks the plane bits up)
    bool frame_initialised = false;
    bool built = false;
};

// the grid walk's own record of a top-level cell (tlg_kernels.hip, k_tlg_trace_records)
struct TlgTopRec { float fres[3]; uint32_t first_leaf; float rcp[3]; uint32_t packed; };
// per plane index of an axis and camera position (tlg_kernels.hip, the packet walk of primary rays): what the leaf-level set-up takes from a
// top-level cell's planes, with the subtractions of the origin done once per camera -- lo - o and hi - o as the walk computes them -- and
// whether lo <= o <= hi (the walk's "origin inside the cell" is the AND of the three axes' flags)
// (near - o, far - o) of a wave with direction sign s on the axis are the two words at byte 16 + 4 * s: { lo - o, hi - o } for d > 0, { hi - o, lo - o } for d < 0
struct TlgCamPlane { float lo, rcp_w, w; uint32_t inside; float lo_o, hi_o, lo_o_again; uint32_t pad; };
static_assert(sizeof(TlgCamPlane) == 32, "TlgCamPlane");
static_assert(sizeof(TlgTopRec) == 32, "TlgTopRec");
struct rt_tlg {
    rt_ctx* ctx = nullptr;
    const rt_scene* scene = nullptr;
    float top_density = 2.0f, leaf_density = 2.0f;    // TwoLevelGridManager.cpp:66-67
    rt_grid_data grid{};
    rt_tlg_info info{};
    uint32_t n = 0;
    uint32_t* d_counters = nullptr;     // max(n, cells, pairs) words
    uint32_t* d_scan = nullptr;
    uint32_t* d_scan_scratch = nullptr;
    uint64_t counters_cap = 0;
    uint2* d_cell_ranges = nullptr;     // cells
    rt_top_cell* d_top = nullptr;       // cells
    uint2* d_pairs = nullptr;           // pairs_pow2
    uint2* d_leaf_pairs = nullptr;      // leaf_pairs_pow2
    uint2* d_leaf_ranges = nullptr;     // leaf_cells_count
    uint32_t* d_written = nullptr;      // 64 words: [0] leaf pairs written; [4..7] counts of a rebuild without round trips (pairs, leaf cells, leaf pairs)
    uint32_t leaf_ranges_cap = 0;       // entries d_leaf_ranges holds
    uint32_t pairs_cap = 0, leaf_cap = 0;   // elements d_pairs / d_leaf_pairs hold (kept across initializeFrame: a rebuild starts on them)
    uint32_t cells_cap = 0;                 // entries d_cell_ranges / d_top hold (the `cells` of the initializeFrame that allocated them)
    uint32_t rebuilds_without_readback = 0;   // diagnostics
    TlgTopRec* d_toprec = nullptr;      // cells_cap records of the walk (null: not allocated, the walk of round 4)
    float4* d_axis_tab = nullptr;       // axis_cap plane records
    uint4* d_leafrec = nullptr;         // leafrec_cap leaf-cell records of the walk
    uint32_t leafrec_cap = 0;
    uint32_t axis_cap = 0;
    bool walk_records = false;          // this build's records are usable (moderate box, no leaf grid finer than 1023 per axis)
    TriCam* d_tricam = nullptr;         // n camera-relative triangle records of the primary-ray trace (rtmi_internal.h, TriCam)
    float tricam_origin[3] = { 0, 0, 0 };
    bool tricam_valid = false;
    uint32_t tricam_n = 0;              // records d_tricam holds
    bool frame_initialised = false;
    bool built = false;
};

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
int rt_fail(const rt_ctx* ctx, const char* file, int line, const char* func, const std::string& msg);
extern "C" __attribute__((visibility("hidden"))) int rt_check_camera(rt_ctx* ctx, const rt_camera* cam);
#define RT_FAIL(ctx, msg) rt_fail((ctx), __FILE__, __LINE__, __func__, (msg))
#define RT_HIP(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return RT_FAIL((ctx), std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

// Entry of every API call that launches work: select the device and drop whatever error an earlier runtime call of this
// thread left behind (hipGetLastError is per thread and sticky, and the host application -- torch, RCCL -- makes calls of its
// own between ours: a stale "invalid resource handle" of theirs was once reported as a failed trace launch).
#define RT_ENTER(ctx)                                                                           \
    do {                                                                                        \
        RT_HIP((ctx), hipSetDevice((ctx)->device));                                             \
        (void)hipGetLastError();                                                                \
    } while (0)

// ---------------------------------------------------------------------------------------
// kernel launchers (one translation unit per family)
// ---------------------------------------------------------------------------------------
// build_kernels.hip
void launch_gather_triangles(hipStream_t s, const DeviceScene& sc);
// leaves + Morton pairs (+ the padding pairs [n, padded) of the pair buffer); on request the search tree's leaves and keys as well
void launch_morton_leaves(hipStream_t s, const DeviceScene& sc, rt_bvh_node* nodes, uint2* pairs, uint32_t padded,
                          rt_bvh_node* s_nodes = nullptr, uint2* s_pairs = nullptr, uint32_t s_padded = 0u,
                          uint32_t* clear = nullptr, uint32_t* s_clear = nullptr,       // + n words each zeroed on the way (the climbs' arrival counters)
                          bool adaptive_keys = true);   // the search tree's key bits allocated by the box's shape (false: ten per axis)
// The kernels of a build carry BOTH hierarchies (the reference tree and the search tree: same leaves, same n) in one launch -- the arguments with a
// trailing 2 are the second one's, blockIdx.y = 1; null: one hierarchy.  A build is a chain of dependent launches of ~5 us each whatever they do.
void launch_radix_tree(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, uint32_t n, uint32_t built, uint2* spans = nullptr, uint32_t* clear2 = nullptr,   // built = n-1 (FIXED) / n-2 (LITERAL_ZERO); clear2: two words zeroed on the way
                       rt_bvh_node* nodes2 = nullptr, const uint2* pairs2 = nullptr, uint2* spans2 = nullptr, uint32_t* clear2_2 = nullptr);
size_t refit_rq_scratch_bytes(uint32_t n);
void launch_refit_rq(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, const uint2* spans, uint32_t n, void* scratch, uint32_t* depth_out,
                     uint32_t* nan_flag_out);      // range-query refit (large scenes); *nan_flag_out != 0: NaN boxes, redo with launch_refit
size_t refit_span_scratch_bytes(uint32_t n);
void launch_refit_span(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, const uint2* spans, uint32_t n, void* scratch, uint32_t* depth_out,
                       uint32_t* nan_flag_out, bool tight,   // refit by spans over a segment tree of the sorted leaf boxes (every size); *nan_flag_out != 0: redo with launch_refit
                       rt_bvh_node* nodes2 = nullptr, const uint2* pairs2 = nullptr, const uint2* spans2 = nullptr, void* scratch2 = nullptr,
                       uint32_t* depth_out2 = nullptr, uint32_t* nan_flag_out2 = nullptr, bool tight2 = false,
                       bool depth_later = false);     // depth_later: the depth comes out of launch_finish (span_scratch there)
void launch_refit_literal(hipStream_t s, rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t* depth_out);
void launch_refit(hipStream_t s, rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t* depth_out, bool tight = false, const uint32_t* only_if = nullptr);   // boxes + tree depth; tight: plain unions (search tree); only_if: runs only when that device word is non-zero
void launch_refit_pair(hipStream_t s, uint32_t n, rt_bvh_node* nodes, uint32_t* counters, uint32_t* depth_out, const uint32_t* only_if,
                       rt_bvh_node* s_nodes, uint32_t* s_counters, uint32_t* s_depth_out, const uint32_t* s_only_if, bool s_tight);   // the climbs of both hierarchies
// the end of a build in one launch: traversal records + ranks + packed triangle records, four-wide records (quad != null), tree depth out of the
// span refit's marks (span_scratch != null), and the depth / flag words (`words`, two per hierarchy) written to `host_words` as well
void launch_finish(hipStream_t s, uint32_t n, bool quad_by_area, const DeviceScene& sc, TriRec* trirec,
                   const rt_bvh_node* nodes, const uint2* pairs, TravNode* trav, uint32_t* pos, uint32_t id_offset, QuadNode* quad, void* span_scratch,
                   uint32_t* words, uint32_t* host_words,
                   const rt_bvh_node* nodes2 = nullptr, const uint2* pairs2 = nullptr, TravNode* trav2 = nullptr, uint32_t* pos2 = nullptr, uint32_t id_offset2 = 0,
                   QuadNode* quad2 = nullptr, void* span_scratch2 = nullptr, uint32_t* words2 = nullptr, uint32_t* host_words2 = nullptr);
// sort_scan_kernels.hip
void launch_bitonic_sort(hipStream_t s, uint2* pairs, uint32_t n_pow2, uint2* pairs2 = nullptr);   // (+ a second array of the same size in the same launches)
size_t radix_sort_scratch_words(uint32_t n);
void launch_radix_sort(hipStream_t s, uint2* pairs, uint2* tmp, uint32_t n, uint32_t key_bits, uint32_t* scratch,   // stable LSD radix sort by the low key_bits of .x
                       uint2* pairs2 = nullptr, uint2* tmp2 = nullptr, uint32_t* scratch2 = nullptr);       // (+ a second sort of the same size in the same launches)
void launch_plane_keys(hipStream_t s, const rt_bvh_node* leaves, uint32_t n, uint32_t axis, uint2* pairs, uint32_t n_pow2);   // sorted (key, .) of the 2n corner coordinates of one axis
int  launch_scan_inclusive(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch);
size_t scan_scratch_words(uint32_t n);
// trace_kernels.hip
// What k_prepare_camera writes per camera position (a strategy asks for what it reads; the rest is not written):
#define RT_PREP_MARKS   1u          // "camera inside child box" bits of the binary records of both trees + the camera-on-a-plane bits
#define RT_PREP_GENERAL 2u          // the camera-relative copy of the four-wide records (general box test)
#define RT_PREP_OCTANT(o) (4u << (o))   // ... with the near / far planes in place for direction octant o
#define RT_PREP_OCTANTS(mask) (((mask) & 0xFFu) << 2)
#define RT_PREP_TRIANGLES 0x400u    // the camera-relative triangle records (k_prepare_triangles)
// Kernel nodes of a captured frame, by role: a frame graph whose camera moves re-sets their arguments
// (hipGraphExecKernelNodeSetParams) instead of being captured again.  The launchers record a node while its launch is captured
// (TraceLaunch::graph_record) and, in update mode (TraceLaunch::graph_update), compute the arguments exactly as for a launch
// and set them on the recorded node instead of launching.
enum { RT_NODE_PREP_CAMERA = 0, RT_NODE_TRACE, RT_NODE_LONG, RT_NODE_RESOLVE, RT_NODE_SHADE, RT_NODE_KINDS };
struct GraphNodes {
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t node[RT_NODE_KINDS] = {};
    bool failed = false;          // a node was missing, appeared twice, or the runtime refused the new arguments
    bool args_only = false;       // update mode: only the argument copies below are refreshed (the sequence is issued as plain launches: the executable graph is not touched)
    // the same launches as plain data (kernel, geometry, a copy of the argument block): a captured sequence that consists of kernels
    // only can be issued as ordinary launches, which follow each other on a stream with a shorter gap than graph launches do
    struct Direct { const void* func = nullptr; dim3 grid, block; uint32_t lds = 0; uint32_t bytes = 0; alignas(16) unsigned char args[768]; } direct[RT_NODE_KINDS];
};
template <typename K, typename A>
inline void rt_launch_node(GraphNodes* record, GraphNodes* update, int kind, K kernel, dim3 grid, dim3 block, hipStream_t s, A& args, uint32_t lds_bytes = 0)
{
    static_assert(sizeof(A) <= sizeof(GraphNodes::Direct::args), "argument block larger than GraphNodes::Direct::args");
    if (GraphNodes* n = update ? update : record) {
        GraphNodes::Direct& d = n->direct[kind];
        d.func = (const void*)kernel; d.grid = grid; d.block = block; d.lds = lds_bytes; d.bytes = (uint32_t)sizeof(A);
        std::memcpy(d.args, &args, sizeof(A));
    }
    if (update && update->args_only) return;
    if (update) {
        hipKernelNodeParams p{};
        void* params[1] = { (void*)&args };
        p.func = (void*)kernel; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = lds_bytes; p.kernelParams = params; p.extra = nullptr;
        if (!update->node[kind] || hipGraphExecKernelNodeSetParams(update->exec, update->node[kind], &p) != hipSuccess) update->failed = true;
        return;
    }
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, s, args);
    if (record) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const hipGraphNode_t* deps = nullptr; size_t n_deps = 0;
        if (hipStreamGetCaptureInfo_v2(s, &st, nullptr, nullptr, &deps, &n_deps) != hipSuccess || st != hipStreamCaptureStatusActive || n_deps != 1 ||
            record->node[kind]) record->failed = true;
        else record->node[kind] = deps[0];
    }
}
struct TraceLaunch {
    bool device_arith;            // the BVH was created with RT_MODE_DEVICE_ARITH: the launchers of the second compilation of trace_kernels.hip
    GraphNodes* graph_record;     // capture in progress: kernel nodes are noted here
    GraphNodes* graph_update;     // no launches: the arguments a launch would get are set on these nodes
    uint32_t prepare_parts;       // RT_PREP_* to (re)compute before the trace (0: nothing)
    uint32_t octant_mask;         // primary rays, four-wide kernel on per-octant records: the octant copies that are valid
    uint32_t plane_bits_known;    // 0x80000000 | bits: the host knows the camera-on-a-plane bits (else the kernels read tree_select)
    const uint32_t* block_order;  // four-wide kernel: wave -> block permutation (null: identity)
    uint32_t waves_per_block;     // first-phase kernels: waves per workgroup (0: the default of four)
    const uint2* tile_xy;         // primary frames: pixel origin of this rank's tile `ordinal` for this (resolution, rank, tiling), or null
    uint32_t* block_cost;         // ... and where it counts this frame's steps per block (null: not counted)
    unsigned long long cal_deadline;   // calibration launches (tag 1): device clock value after which the kernels stop walking, 0 = none
    const TravNode* trav;         // reference tree records [0, n-1), then (optionally) the search tree's [n-1, 2n-2)
    const QuadNode* quad;         // four-wide records (null: the packet kernel walks the binary records)
    const QuadNode* quad_cam;     // primary rays: the same records with the camera position subtracted from the boxes (+ thresholds)
    uint32_t quad_cam_variants;   // 1: the general copy only; 9: + one per direction octant (near / far planes in place)
    bool no_unsafe_rays;          // the caller knows that the camera sits on no box plane: the four-wide kernel hands no ray over
    // Fused frame (frame graphs, rt_render_frame): the kernels shade every contact as they store it (shade_math.h) -- asked for by
    // fused_rgba != null, honoured (*fused_done = true) by the four-wide kernel and by the lane-per-ray kernel, each together with the
    // wave-per-ray kernel behind it (a ray is stored by exactly one of the two); the binary packet kernel leaves the colours to k_shade.
    // plane stride != 0: three float planes instead of float4 (shade_kernels.hip)
    void* fused_rgba;
    uint32_t fused_plane_stride;
    bool* fused_done;
    uint32_t search_root;         // primary rays: root record of the search tree, 0 = none (walk the reference tree)
    const uint32_t* tree_select;  // per camera: bit k set = the camera's coordinate k equals a box plane
    const DeviceScene* scene;
    const uint32_t* pos;          // leaf -> position in the sorted pair array (DFS rank, tie rule)
    const TriRec* trirec;         // packed triangle records (packet kernels)
    const TriCam* tricam;         // primary rays, four-wide packet kernel: the same relative to the camera position (may be null)
    uint32_t n_leaves;
    void* ctrl;                   // TraceCtrl, zeroed by the launcher
    uint32_t* long_list;          // records of rays deferred to the wave-per-ray phase
    uint32_t long_cap;            // records available
    uint32_t defer_budget;        // 0 = default
    hipEvent_t ev[3];             // optional: recorded before phase 0, between the phases, after phase 1
    bool diagnostics;             // knob RTMI_TRACE_DIAG=1: the wave-per-ray kernel counts its pool steps
    int tag;                      // kernel-name tag: 0 steady state, 1 first-frame calibration, 2 isolated measurement pass
                                  // (rt_ctx_set_profiling(ctx, 2)); same code, separate names in profiles
    bool packet4;                 // with packet: walk the four-wide records
    bool packet;                  // primary rays: packet kernel (one 8x8 block per wave) instead of the two-phase kernels
    uint32_t* packet_spill;       // bitmap + overflow areas of the packet stacks (null when the tree is shallow)
    unsigned long long* keys;     // per work item, split rays only
    uint32_t* split_list;         // long_cap entries
    const uint2* pairs;           // sorted (morton, leaf): position -> leaf
    uint32_t packet_spill_entries;
    uint32_t packet_spill_slots;  // areas behind the bitmap at the start of packet_spill
    int compute_units;
    void* task_shards;            // k_trace_packet4: queues of sub-tree tasks shared between waves (trace_task_bytes(task_cap)), may be null
    uint32_t task_cap;            // entries per queue
    uint32_t task_helper_blocks;  // 0 = default
    uint32_t long_batch;          // rays per wave of the wave-per-ray kernel, 0 = by the number of deferred rays
};
void launch_trace_primary(hipStream_t s, const TraceLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks,
                          bool compact, rt_contact* out);
void launch_trace_rays(hipStream_t s, const TraceLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out);
// the same three from the second compilation of trace_kernels.hip (RT_MODE_DEVICE_ARITH: dot / cross as ROCm's OpenCL C library fuses them)
void launch_trace_primary_devarith(hipStream_t s, const TraceLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks, bool compact, rt_contact* out);
void launch_trace_rays_devarith(hipStream_t s, const TraceLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out);
bool launch_prepare_camera_devarith(hipStream_t s, const TraceLaunch& t, const float origin[3], void* clear_a = nullptr, size_t clear_a_bytes = 0, void* clear_b = nullptr, size_t clear_b_bytes = 0);
bool rt_preload_trace_devarith();
bool launch_prepare_camera(hipStream_t s, const TraceLaunch& t, const float origin[3], void* clear_a = nullptr, size_t clear_a_bytes = 0,   // t.prepare_parts; true: launched
                           void* clear_b = nullptr, size_t clear_b_bytes = 0);     // (+ two buffers zeroed on the way: an eager frame's fills)
void launch_device_clock(hipStream_t s, unsigned long long* d_out);      // the clock the calibration deadlines are on (s_memrealtime)
bool launch_block_order(hipStream_t s, const uint32_t* cost, uint32_t* order, uint32_t blocks, uint32_t group_log, bool identity);
bool rt_preload_trace(); bool rt_preload_build(); bool rt_preload_secondary(); bool rt_preload_shade(); bool rt_preload_sort_scan(); bool rt_preload_tlg();
size_t trace_ctrl_bytes();
size_t trace_task_bytes(uint32_t cap_per_shard);
size_t trace_ctrl_error_offset();
size_t trace_ctrl_expired_offset();
void trace_ctrl_counter_offsets(size_t out[RT_TRACE_COUNTERS]);
size_t trace_long_record_bytes();
uint32_t trace_packet_stack_entries();
uint32_t trace_lds_stack_entries();
// tlg_kernels.hip
struct TlgLaunch {
    rt_grid_data grid;
    const rt_top_cell* top;
    const uint2* leaf_ranges;
    const uint2* leaf_pairs;
    const DeviceScene* scene;
    const TriCam* tricam;      // primary rays: the triangle records relative to the camera position (null: the scene's table)
    bool prepare_tricam;       // ... which this launch (re)computes first: the camera moved, or the scene was rebuilt
    const TlgTopRec* toprec;   // the walk's records (null: none, the walk of round 4)
    const float4* axis_tab;
    const uint4* leafrec;
    TlgCamPlane* axis_cam;     // primary rays with the records: the planes relative to the camera position, (re)computed with tricam; they live behind
                               // the cell records in toprec's buffer (one base pointer for both in the walk), at byte cam_off
    uint32_t cam_off;
};
void launch_tlg_leaf_records(hipStream_t s, const uint2* leaf_ranges, const uint2* leaf_pairs, uint32_t leaf_cells, uint32_t pairs_cap, uint4* rec);
void launch_tlg_trace_records(hipStream_t s, const rt_grid_data& g, const rt_top_cell* top, uint32_t cells, TlgTopRec* rec, float4* axis_tab, uint32_t* too_fine);
void launch_tlg_count_top(hipStream_t s, const rt_grid_data& g, const DeviceScene& sc, uint32_t* counters);
void launch_tlg_emit_top(hipStream_t s, const rt_grid_data& g, const DeviceScene& sc, const uint32_t* scan, const uint32_t* counters, uint2* pairs,
                         uint32_t pairs_cap);
// count_dev != null: the real element count is read on the device and `count` is the capacity the grid covers
void launch_segment_bounds(hipStream_t s, const uint2* sorted, uint32_t count, const uint32_t* count_dev, uint2* ranges, uint32_t ranges_cap);
void launch_tlg_cell_res(hipStream_t s, const rt_grid_data& g, const uint2* ranges, uint32_t cells, rt_top_cell* top, uint32_t* leaves);
void launch_tlg_first_leaf(hipStream_t s, rt_top_cell* top, const uint32_t* scan, uint32_t cells);
void launch_tlg_count_leaf(hipStream_t s, const rt_grid_data& g, const uint2* pairs, uint32_t pairs_count, const uint32_t* count_dev,
                           const rt_top_cell* top, const DeviceScene& sc, uint32_t* counters);
void launch_tlg_emit_leaf(hipStream_t s, const rt_grid_data& g, const uint2* pairs, uint32_t pairs_count, const uint32_t* count_dev,
                          const rt_top_cell* top, const DeviceScene& sc, const uint32_t* scan, const uint32_t* counters, uint2* leaf_pairs,
                          uint32_t leaf_cap, uint32_t* written_total);
void launch_tlg_trace_primary(hipStream_t s, const TlgLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks, bool compact,
                              rt_contact* out, float4* fused_rgba = nullptr);      // fused_rgba: the frame's colours as well (untiled frames)
void launch_tlg_trace_rays(hipStream_t s, const TlgLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out);
// secondary_kernels.hip
void launch_secondary_flags(hipStream_t s, const rt_contact* contacts, uint32_t count, uint32_t* flags);
void launch_primary_rays(hipStream_t s, const rt_camera* cam, rt_ray* out);
void launch_secondary_emit(hipStream_t s, const rt_camera* cam, const rt_ray* rays_in, const rt_contact* contacts, uint32_t count,
                           const uint32_t* scan, int kind, const float target[3], float eps, rt_ray* out);
// shade_kernels.hip
void launch_shade(hipStream_t s, const DeviceScene& sc, const rt_camera& cam, const rt_contact* contacts,
                  uint32_t count, float4* rgba, bool tiled, uint32_t rank, uint32_t nranks, uint32_t plane_stride = 0,
                  GraphNodes* graph_record = nullptr, GraphNodes* graph_update = nullptr);
void launch_untile_rgba(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float4* gathered,
                        uint32_t slots_per_rank, float4* frame);
void launch_untile_rgba_packed(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float* gathered,
                               uint32_t slots_per_rank, uint64_t rank_stride, float4* frame);

void launch_untile_rgba_split(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float* root_planes, uint32_t root_slots,
                              const float* peers, uint32_t peer_slots, uint64_t peer_stride, float4* frame);

// tiling helpers shared by host and device (64x64-pixel tiles dealt round-robin)
#define RT_TILE 64u
__host__ __device__ inline uint32_t rt_tiles_x(uint32_t resX) { return (resX + RT_TILE - 1) / RT_TILE; }
__host__ __device__ inline uint32_t rt_tiles_y(uint32_t resY) { return (resY + RT_TILE - 1) / RT_TILE; }
// Who owns which tile.  `nranks` is the word of the C ABI: low 16 bits = number of ranks N, high 16 bits = e, the extra tiles
// per period of N + e tiles that go to rank 0 (RT_TILING in rtmi.h; e = 0 is plain round robin).  With e > 0 the rank that
// gathers the frame traces more of it itself and receives less over its links.
//   position p = tile % (N + e) in its period q = tile / (N + e):  p <= e -> rank 0, ordinal q (e + 1) + p;  else rank p - e, ordinal q
__host__ __device__ inline uint32_t rt_nranks(uint32_t nranks_word) { return nranks_word & 0xFFFFu; }
__host__ __device__ inline uint32_t rt_root_extra(uint32_t nranks_word) { return nranks_word >> 16; }
__host__ __device__ inline void rt_tile_owner(uint32_t tile, uint32_t nranks_word, uint32_t& rank, uint32_t& ordinal)
{
    const uint32_t e = rt_root_extra(nranks_word), period = rt_nranks(nranks_word) + e;
    const uint32_t q = tile / period, p = tile - q * period;
    if (p <= e) { rank = 0u; ordinal = q * (e + 1u) + p; }
    else { rank = p - e; ordinal = q; }
}
__host__ __device__ inline uint32_t rt_tile_of(uint32_t rank, uint32_t ordinal, uint32_t nranks_word)
{
    const uint32_t e = rt_root_extra(nranks_word), period = rt_nranks(nranks_word) + e;
    if (rank == 0u) { const uint32_t q = ordinal / (e + 1u); return q * period + (ordinal - q * (e + 1u)); }
    return ordinal * period + rank + e;
}
__host__ __device__ inline uint32_t rt_owned_tiles(uint32_t total_tiles, uint32_t rank, uint32_t nranks_word)
{
    const uint32_t e = rt_root_extra(nranks_word), period = rt_nranks(nranks_word) + e;
    const uint32_t q = total_tiles / period, rem = total_tiles - q * period;
    if (rank == 0u) return q * (e + 1u) + (rem < e + 1u ? rem : e + 1u);
    return q + (rem > rank + e ? 1u : 0u);
}
