// trace_kernels.hip -- the hot path: BVH traversal + ray/AABB + ray/triangle for gfx950.
//
// Semantics are those of bvh_generate_contact (Include/CLData/AccelerationStructs/BVH.h:344-408):
// iterative depth-first walk, child A first / child B pushed, NO closest-hit pruning, a child is
// entered iff (AABBIntersect > 0 || origin inside the box) (BVH.h:366-369, AABB.h:126-142,178-184),
// the closest hit is kept with a strict `<` (BVH.h:392), so among equal-t hits the first leaf in
// DFS order wins.  The arithmetic is restated operation by operation (device_math.h).
//
// What is different is the machine mapping (MI355X, wave64):
//  * persistent waves: a fixed grid sized to the chip pulls rays from a global work counter;
//    a lane whose ray has finished is refilled with a fresh ray (ballot + prefix popcount
//    packing) as soon as enough lanes of its wave are idle, instead of idling until the
//    slowest ray of a 64-ray work-group is done (reference: one work-item per ray, WG = 64,
//    BVHManager.cpp:264-270);
//  * the traversal stack lives in LDS, one private column per lane ([level][lane] => no bank
//    conflicts); only trees deeper than the LDS part spill to a per-lane global area;
//  * one inner-node visit = one 64-byte record holding both child boxes and both child ids
//    (TravNode) instead of a 48-byte node + two dependent 32-byte box fetches;
//  * leaves read an SoA triangle table (v0, e1, e2 as float4) instead of chasing
//    ModelHeader/MeshHeader.dataSize chains and ushort indices (SceneBufferParser.h:175-217);
//    edges and the geometric normal are ray independent and are precomputed with the same
//    operations (build_kernels.hip);
//  * leaf tests are postponed: a lane parks one reached leaf and keeps walking inner nodes; the
//    wave runs the triangle test when a lane needs its slot again or is about to finish, so the
//    triangle code is not executed on every iteration.  Per lane the leaves are still tested in
//    DFS order, which is all the tie rule needs;
//  * pixels are walked in 8x8 blocks inside 64x64 tiles, so a wave's rays are neighbours.
#include "rtmi_internal.h"
#include "device_math.h"

#define TRACE_BLOCK 256u
#define TRACE_WAVES (TRACE_BLOCK / 64u)
#define LDS_STACK 24u            // entries per lane kept in LDS (observed need at 1080p: <= 18)
#define NO_LEAF 0xFFFFFFFEu
#define REFILL_MIN_IDLE 16       // refill a wave once this many lanes are idle

uint32_t trace_lds_stack_entries() { return LDS_STACK; }

struct TraceArgs {
    const TravNode* __restrict__ trav;
    const float4* __restrict__ v0;
    const float4* __restrict__ e1;
    const float4* __restrict__ e2;
    const float4* __restrict__ nrm;
    uint32_t* work_counter;
    uint32_t* overflow;
    uint32_t overflow_words_per_lane;
    uint32_t total_work;
    // primary rays
    rt_camera cam;
    uint32_t rank, nranks, tiles_x, total_tiles;
    uint32_t compact;
    // ray buffer
    const rt_ray* __restrict__ rays;
    rt_contact* __restrict__ out;
};

struct RayState {
    f3 o, d, inv;
    float best_t;
    uint32_t best_leaf;
    uint32_t out_index;      // where the contact is stored
    uint32_t pixel;          // Contact.pixelIndex
};

// one child of an inner node: AABBIntersect (AABB.h:126-142) folded with the validity rule of
// BVH.h:368-369.  `tmin*flag > 0` <=> `flag && tmin > 0` (x*1.0f == x; x*0.0f is +-0 or NaN).
__device__ __forceinline__ bool child_valid(const float lo[3], const float hi[3], const RayState& r,
                                            bool sx, bool sy, bool sz)
{
    float tmin  = ((sx ? hi[0] : lo[0]) - r.o.x) * r.inv.x;
    float tmax  = ((sx ? lo[0] : hi[0]) - r.o.x) * r.inv.x;
    float tymin = ((sy ? hi[1] : lo[1]) - r.o.y) * r.inv.y;
    float tymax = ((sy ? lo[1] : hi[1]) - r.o.y) * r.inv.y;
    bool flag = !((tmin > tymax) || (tymin > tmax));
    tmin = RMAX(tmin, tymin);
    tmax = RMIN(tmax, tymax);
    float tzmin = ((sz ? hi[2] : lo[2]) - r.o.z) * r.inv.z;
    float tzmax = ((sz ? lo[2] : hi[2]) - r.o.z) * r.inv.z;
    flag = flag && !((tmin > tzmax) || (tzmin > tmax));
    tmin = RMAX(tmin, tzmin);
    bool inside = (lo[0] <= r.o.x) && (r.o.x <= hi[0]) && (lo[1] <= r.o.y) && (r.o.y <= hi[1]) &&
                  (lo[2] <= r.o.z) && (r.o.z <= hi[2]);
    return (flag && (tmin > 0.0f)) || inside;
}

// triangleIntersect, Triangle.h:66-107, against precomputed edges; keeps (t, leaf) under BVH.h:392.
__device__ __forceinline__ void leaf_test(const TraceArgs& a, uint32_t leaf, RayState& r)
{
    float4 v0 = a.v0[leaf], E1 = a.e1[leaf], E2 = a.e2[leaf];
    f3 vert0 = f3_from4(v0), edge1 = f3_from4(E1), edge2 = f3_from4(E2);
    f3 pvec = f3_cross(r.d, edge2);
    float det = f3_dot(edge1, pvec);
    bool flag = fabsf(det) >= FLT_EPSILON;
    float inv_det = 1.0f / det;
    f3 tvec = f3_sub(r.o, vert0);
    float u = f3_dot(tvec, pvec) * inv_det;
    flag = flag && (u < 1.0f);
    flag = flag && (u >= 0.0f);
    f3 qvec = f3_cross(tvec, edge1);
    float v = f3_dot(r.d, qvec) * inv_det;
    flag = flag && ((u + v) <= 1.0f);
    flag = flag && (v >= 0.0f);
    float t = f3_dot(edge2, qvec) * inv_det;      // "* flag" folded into the test below
    if (flag && t > 0.0f && t < r.best_t) { r.best_t = t; r.best_leaf = leaf; }
}

// work item -> pixel.  Work items enumerate this rank's 64x64 tiles; inside a tile the 64 8x8
// blocks, inside a block the 64 pixels (one wave refill batch ~ one 8x8 block).
__device__ __forceinline__ bool work_to_pixel(const TraceArgs& a, uint32_t w, uint32_t& pixel, uint32_t& slot)
{
    uint32_t ordinal = w >> 12;                 // owned tile ordinal (4096 px per tile)
    uint32_t within = w & 4095u;
    uint32_t tile = ordinal * a.nranks + a.rank;
    uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    uint32_t blk = within >> 6, lp = within & 63u;
    uint32_t ly = ((blk >> 3) << 3) + (lp >> 3);
    uint32_t lx = ((blk & 7u) << 3) + (lp & 7u);
    uint32_t x = tx * RT_TILE + lx, y = ty * RT_TILE + ly;
    slot = (ordinal << 12) + ly * RT_TILE + lx;
    pixel = y * a.cam.resX + x;
    return x < a.cam.resX && y < a.cam.resY;
}

template <int PRIMARY>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(TraceArgs a)
{
    __shared__ uint32_t s_stack[TRACE_WAVES][LDS_STACK][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t (*stack)[64] = s_stack[wave];
    uint32_t* spill = a.overflow
        ? a.overflow + (uint64_t)(blockIdx.x * TRACE_BLOCK + threadIdx.x) * a.overflow_words_per_lane
        : nullptr;

    RayState r;
    r.o = f3{ 0, 0, 0 }; r.d = f3{ 0, 0, 1 }; r.inv = f3{ 0, 0, 1 };
    r.best_t = FLT_MAX; r.best_leaf = NO_LEAF; r.out_index = 0; r.pixel = 0;
    bool sx = false, sy = false, sz = false;
    uint32_t cur = RT_SENTINEL;       // node to visit: inner record index, RT_LEAF_FLAG|leaf, or sentinel
    uint32_t pending = NO_LEAF;       // parked leaf
    uint32_t sp = 0;                  // entries on this lane's stack
    bool active = false;
    bool exhausted = false;           // wave-uniform: the work counter ran dry

    for (;;) {
        // ---------------- refill idle lanes (wavefront ballot / compaction) ----------------
        uint64_t idle = __ballot(!active);
        if (idle != 0ull && !exhausted) {
            uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle >= REFILL_MIN_IDLE || n_idle == 64u) {
                uint32_t base = 0;
                int leader = __ffsll((unsigned long long)idle) - 1;
                if ((int)lane == leader) base = atomicAdd(a.work_counter, n_idle);
                base = (uint32_t)__shfl((int)base, leader, 64);
                if (base + n_idle >= a.total_work) exhausted = true;
                if (!active) {
                    uint32_t w = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    if (w < a.total_work) {
                        bool ok = true;
                        if (PRIMARY) {
                            uint32_t pixel, slot;
                            ok = work_to_pixel(a, w, pixel, slot);
                            if (ok) {
                                generate_ray(a.cam, pixel, r.o, r.d);
                                r.pixel = pixel;
                                r.out_index = a.compact ? slot : pixel;
                            } else if (a.compact) {
                                // slot of a border tile that lies outside the frame: defined as "no contact"
                                float4* o = reinterpret_cast<float4*>(a.out + slot);
                                o[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                o[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            }
                        } else {
                            // kernel generateContacts2, BVHKernels.cl:103-117
                            const rt_ray ry = a.rays[w];
                            r.o = f3{ ry.origin[0], ry.origin[1], ry.origin[2] };
                            r.d = f3{ ry.direction[0], ry.direction[1], ry.direction[2] };
                            r.pixel = w; r.out_index = w;
                        }
                        if (ok) {
                            r.inv = f3{ 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };   // INV_DIR_*, AABB.h:115-117
                            sx = 0.0f > r.inv.x; sy = 0.0f > r.inv.y; sz = 0.0f > r.inv.z;
                            r.best_t = FLT_MAX; r.best_leaf = NO_LEAF;
                            cur = 0u;              // root = inner record 0 (reference node index n)
                            sp = 0; pending = NO_LEAF;
                            active = true;
                        }
                    }
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;                 // every pulled slot was outside the frame: pull again
        }

        // ---------------- traversal steps until enough lanes have drained ----------------
        for (;;) {
            if (active) {
                if (cur != RT_SENTINEL && (cur & RT_LEAF_FLAG) == 0u) {
                    // inner node: one 64-byte record
                    const float4* p = reinterpret_cast<const float4*>(a.trav + cur);
                    float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                    float alo[3] = { q0.x, q0.y, q0.z }, ahi[3] = { q1.x, q1.y, q1.z };
                    float blo[3] = { q2.x, q2.y, q2.z }, bhi[3] = { q3.x, q3.y, q3.z };
                    uint32_t ca = __float_as_uint(q0.w), cb = __float_as_uint(q1.w);
                    bool aValid = child_valid(alo, ahi, r, sx, sy, sz);
                    bool bValid = child_valid(blo, bhi, r, sx, sy, sz);
                    if (aValid && bValid) {
                        cur = ca;
                        if (sp < LDS_STACK) stack[sp][lane] = cb; else spill[sp - LDS_STACK] = cb;
                        sp++;
                    } else if (aValid) cur = ca;
                    else if (bValid) cur = cb;
                    else if (sp == 0) cur = RT_SENTINEL;
                    else { sp--; cur = (sp < LDS_STACK) ? stack[sp][lane] : spill[sp - LDS_STACK]; }
                }
                // park a reached leaf if the slot is free, and move on
                if (cur != RT_SENTINEL && (cur & RT_LEAF_FLAG) != 0u && pending == NO_LEAF) {
                    pending = cur & ~RT_LEAF_FLAG;
                    if (sp == 0) cur = RT_SENTINEL;
                    else { sp--; cur = (sp < LDS_STACK) ? stack[sp][lane] : spill[sp - LDS_STACK]; }
                }
            }
            // a lane that needs its slot again (second leaf) or is about to finish forces the
            // wave through the triangle test for every parked leaf
            bool must = active && pending != NO_LEAF &&
                        (cur == RT_SENTINEL || (cur & RT_LEAF_FLAG) != 0u);
            if (__ballot(must) != 0ull) {
                if (active && pending != NO_LEAF) { leaf_test(a, pending, r); pending = NO_LEAF; }
            }
            // retire finished rays: Contact assembly, BVH.h:402-407 + BVHKernels.cl:95-96
            if (active && cur == RT_SENTINEL && pending == NO_LEAF) {
                float4 nt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                uint32_t material = 0;
                if (r.best_leaf != NO_LEAF) {
                    float4 n = a.nrm[r.best_leaf];
                    nt = make_float4(n.x, n.y, n.z, r.best_t);
                    material = __float_as_uint(a.v0[r.best_leaf].w);
                }
                uint4 head = make_uint4(r.pixel, material, 0u, 0u);
                float4* o = reinterpret_cast<float4*>(a.out + r.out_index);
                o[0] = *reinterpret_cast<float4*>(&head);
                o[1] = nt;
                active = false;
            }
            uint32_t n_active = (uint32_t)__popcll(__ballot(active));
            if (n_active == 0u) break;
            if (!exhausted && n_active <= 64u - REFILL_MIN_IDLE) break;
        }
    }
}

uint32_t trace_max_resident_lanes(int compute_units)
{
    int blocks_per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_trace<1>, TRACE_BLOCK, 0);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    int b0 = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, k_trace<0>, TRACE_BLOCK, 0);
    if (b0 > blocks_per_cu) blocks_per_cu = b0;
    return (uint32_t)compute_units * (uint32_t)blocks_per_cu * TRACE_BLOCK;
}

template <int PRIMARY>
static void launch(hipStream_t s, const TraceLaunch& t, TraceArgs& a)
{
    int blocks_per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_trace<PRIMARY>, TRACE_BLOCK, 0);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    uint32_t grid = (uint32_t)t.compute_units * (uint32_t)blocks_per_cu;
    uint32_t need = (a.total_work + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if (grid > need) grid = need;
    if (grid == 0) return;
    hipMemsetAsync(t.work_counter, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_trace<PRIMARY>, dim3(grid), dim3(TRACE_BLOCK), 0, s, a);
}

static void fill_common(const TraceLaunch& t, TraceArgs& a)
{
    a.trav = t.trav;
    a.v0 = t.scene->v0; a.e1 = t.scene->e1; a.e2 = t.scene->e2; a.nrm = t.scene->nrm;
    a.work_counter = t.work_counter;
    a.overflow = t.overflow;
    a.overflow_words_per_lane = t.overflow_words_per_lane;
}

void launch_trace_primary(hipStream_t s, const TraceLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks,
                          bool compact, rt_contact* out)
{
    TraceArgs a{};
    fill_common(t, a);
    a.cam = cam;
    a.rank = rank; a.nranks = nranks;
    a.tiles_x = rt_tiles_x(cam.resX);
    a.total_tiles = a.tiles_x * rt_tiles_y(cam.resY);
    a.compact = compact ? 1u : 0u;
    a.total_work = rt_owned_tiles(a.total_tiles, rank, nranks) * RT_TILE * RT_TILE;
    a.rays = nullptr;
    a.out = out;
    launch<1>(s, t, a);
}

void launch_trace_rays(hipStream_t s, const TraceLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out)
{
    TraceArgs a{};
    fill_common(t, a);
    a.total_work = count;
    a.rays = rays;
    a.out = out;
    a.nranks = 1;
    launch<0>(s, t, a);
}
