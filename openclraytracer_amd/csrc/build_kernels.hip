// build_kernels.hip -- BVH construction on the device (gfx950).
//
// Restates, as hand-written HIP kernels over an SoA triangle table, the reference's build chain
//   calculateMortonCodes -> BitonicSort -> buildRadixTree -> computeBoundingBoxes
// (src/CLRayTracerEngine/BVHKernels.cl:46-79, Include/CLData/AccelerationStructs/BVH.h:49-334,
//  BVHManager.cpp:156-228) and adds the re-layout into 64-byte traversal records.
// Integer work must be bit-exact with the reference; float work uses device_math.h.
#include "rtmi_internal.h"
#include <cstdlib>
#include "device_math.h"

#include <limits.h>

// ----------------------------------------------------------------------------------------
// 1. scene bytes -> SoA triangle table
//    replaces the per-access pointer chase getModelAtIndex/getMeshAtIndex/getIndexAt/getVertexAt
//    (SceneBufferParser.h:175-217) and getTriangleRefByIndex (:253-293).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float load_f32_unaligned(const unsigned char* p)
{
    // meshes after the first are only 2-byte aligned (mesh size = 48 + 16 nv + 2 ni)
    const uint16_t* h = reinterpret_cast<const uint16_t*>(p);
    uint32_t bits = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    return __uint_as_float(bits);
}

__global__ __launch_bounds__(256) void k_gather_triangles(DeviceScene sc)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sc.n_tris) return;
    // binary search: last mesh with first_triangle <= t (meshes with 0 triangles never match)
    uint32_t lo = 0, hi = sc.n_meshes;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (sc.meshes[mid].first_triangle <= t) lo = mid; else hi = mid;
    }
    while (sc.meshes[lo].triangle_count == 0 || t >= sc.meshes[lo].first_triangle + sc.meshes[lo].triangle_count) lo--;
    const MeshEntry me = sc.meshes[lo];
    uint32_t local = t - me.first_triangle;
    const unsigned char* mesh = sc.bytes + me.byte_offset;
    const unsigned char* verts = mesh + 48;                                   // MESH_HEADER_SIZE
    const uint16_t* idx = reinterpret_cast<const uint16_t*>(verts + (uint64_t)me.vertex_count * 16);
    f3 v[3];
    for (int k = 0; k < 3; k++) {
        // INDEX_TYPE = ushort (SceneBufferParser.h:160); 32-bit indices (extension) are read as two halves: meshes are 2-byte aligned
        const uint32_t e = local * 3 + k;
        uint32_t vi = sc.index_bytes == 4u ? ((uint32_t)idx[2 * e] | ((uint32_t)idx[2 * e + 1] << 16)) : (uint32_t)idx[e];
        const unsigned char* p = verts + (uint64_t)vi * 16;                   // VERTEX_SIZE = 16
        v[k].x = load_f32_unaligned(p);
        v[k].y = load_f32_unaligned(p + 4);
        v[k].z = load_f32_unaligned(p + 8);
    }
    f3 e1 = f3_sub(v[1], v[0]);                                               // Triangle.h:72-73
    f3 e2 = f3_sub(v[2], v[0]);
    f3 n = f3_normalize(f3_cross(e1, e2));                                    // Triangle.h:105
    sc.v0[t] = make_float4(v[0].x, v[0].y, v[0].z, __uint_as_float(me.material_index));
    sc.v1[t] = make_float4(v[1].x, v[1].y, v[1].z, 0.0f);
    sc.v2[t] = make_float4(v[2].x, v[2].y, v[2].z, 0.0f);
    sc.e1[t] = make_float4(e1.x, e1.y, e1.z, 0.0f);
    sc.e2[t] = make_float4(e2.x, e2.y, e2.z, 0.0f);
    sc.nrm[t] = make_float4(n.x, n.y, n.z, 0.0f);
    sc.ref[t] = make_uint4(local, me.submesh_index, me.model_index, me.material_index);
}

void launch_gather_triangles(hipStream_t s, const DeviceScene& sc)
{
    if (!sc.n_tris) return;
    hipLaunchKernelGGL(k_gather_triangles, dim3((sc.n_tris + 255) / 256), dim3(256), 0, s, sc);
}

// ----------------------------------------------------------------------------------------
// 2. Morton codes + leaf nodes: calculateMorton, BVH.h:83-130 (host branch: centroid per
//    component (a+b+c)/3.0f), calculateTriangleAABB AABB.h:61-79, normalizeScale
//    RTKernelUtils.h:63-81 (value/oldLen, oldMin NOT subtracted), morton3D BVH.h:65-73.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(float x, float y, float z)
{
    x = RMIN(RMAX(x * 1024.0f, 0.0f), 1023.0f);
    y = RMIN(RMAX(y * 1024.0f, 0.0f), 1023.0f);
    z = RMIN(RMAX(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((uint32_t)x) * 4 + expand_bits((uint32_t)y) * 2 + expand_bits((uint32_t)z);
}
__device__ __forceinline__ float normalize_scale(float oldMin, float oldMax, float value)
{
    float oldLen = oldMax - oldMin;
    float percentage = value / oldLen;
    float newLen = 1.0f - 0.0f;
    return percentage * newLen + 0.0f;
}

// (threads beyond the last triangle write the pair buffer's padding, (UINT_MAX, UINT_MAX) up to the power of two N: BVHManager.cpp:134-140 clears
//  the whole buffer to it before the kernel; every entry below n is overwritten here anyway, so the clearing pass is this kernel's tail.)
// The same pass feeds the SEARCH tree (section 6 below) when the caller passes its buffers: a copy of the leaf node and the key of the box
// centre normalised to the scene header's box -- both need the triangle and nothing else, so neither a copy of the leaves nor a key kernel
// stands at the head of that hierarchy's chain of launches.
// The search tree's key: 30 bits of the box centre's position in the scene header's box -- not ten per axis: every bit halves the cells along the
// axis on which they are LONGEST at that point (key_layout below), so that the cells of every level are as close to cubes as the box allows.
// ManFigure's box is 16 x 18.5 x 3.4: ten bits per axis cut it into slabs five times thinner than wide, and a radix tree over such keys
// separates what lies next to each other in x and y long before it has to (tools/tree_quality.py's model: -23 % four-wide steps on ManFigure's
// frame, -38 % on Gardenlight's 1.5 x 6.5 x 1.5).  A box whose allocation stays within two bits between its axes keeps the plain ten bits per
// axis, x, y, z interleaved (Cube, ClumsyDragon, Tenryuu: the same keys as before).  The order is the library's own: no contact depends on it.
struct KeyBox { float lo[3], hi[3]; uint32_t nbits[3]; unsigned long long seq; };    // seq: 2 bits per key bit (its axis), the key's top bit first
__device__ __forceinline__ uint32_t search_tree_key(const float lo[3], const float hi[3], const KeyBox& box)
{
    float u[3];
    for (int k = 0; k < 3; k++) {
        float c = (lo[k] + hi[k]) * 0.5f;
        float ext = box.hi[k] - box.lo[k];
        float v = (c - box.lo[k]) / ext;
        u[k] = (v == v) ? v : 0.0f;                      // flat axis: 0/0
    }
    if (box.seq == 0ull) return morton3d(u[0], u[1], u[2]);
    uint32_t q[3], left[3];
    for (int k = 0; k < 3; k++) {
        const uint32_t cells = 1u << box.nbits[k];                      // (at most 2^30: one axis takes every bit of a box that is a line)
        const uint32_t at = (uint32_t)RMIN(RMAX(u[k] * (float)cells, 0.0f), (float)cells);
        q[k] = at < cells - 1u ? at : cells - 1u;                      // (an integer clamp: (float)(cells - 1) rounds up to cells from 2^25 on)
        left[k] = box.nbits[k];
    }
    uint32_t code = 0;
    for (int b = 29; b >= 0; b--) {
        const uint32_t a = (uint32_t)(box.seq >> (2 * b)) & 3u;
        const uint32_t l = (a == 0u ? left[0] : a == 1u ? left[1] : left[2]) - 1u;
        const uint32_t v = a == 0u ? q[0] : a == 1u ? q[1] : q[2];
        left[0] -= a == 0u ? 1u : 0u; left[1] -= a == 1u ? 1u : 0u; left[2] -= a == 2u ? 1u : 0u;
        code = (code << 1) | ((v >> l) & 1u);
    }
    return code;
}
// which axis every bit of the key belongs to (host): the axis along which the current cell is longest, ties to the lower axis
static void key_layout(KeyBox& box)
{
    double cell[3];
    for (int k = 0; k < 3; k++) { const double e = (double)box.hi[k] - (double)box.lo[k]; cell[k] = e > 0.0 ? e : 0.0; box.nbits[k] = 0u; }
    box.seq = 0ull;
    unsigned long long seq = 0ull;
    for (int b = 0; b < 30; b++) {
        int a = 0;
        if (cell[1] > cell[a]) a = 1;
        if (cell[2] > cell[a]) a = 2;
        seq = (seq << 2) | (unsigned long long)a;
        cell[a] *= 0.5; box.nbits[a]++;
    }
    // (a box that is nearly a cube keeps ten bits per axis: an allocation one bit off it reorders the interleave and little else -- Tenryuu,
    //  0.64 x 1.34 x 0.9 -> 9 / 11 / 10, measured 5 % slower than with 10 / 10 / 10)
    const uint32_t most = box.nbits[0] > box.nbits[1] ? (box.nbits[0] > box.nbits[2] ? box.nbits[0] : box.nbits[2]) : (box.nbits[1] > box.nbits[2] ? box.nbits[1] : box.nbits[2]);
    const uint32_t least = box.nbits[0] < box.nbits[1] ? (box.nbits[0] < box.nbits[2] ? box.nbits[0] : box.nbits[2]) : (box.nbits[1] < box.nbits[2] ? box.nbits[1] : box.nbits[2]);
    if (most - least >= 3u) box.seq = seq | (1ull << 62);               // (bit 62: "set", whatever the axes)
}
// (clear / s_clear: the arrival counters of a climbing refit behind this kernel, one word per leaf, zeroed on the way -- BVHManager.cpp:141-147 --
//  instead of by a fill launch of their own)
__global__ __launch_bounds__(256) void k_morton_leaves(DeviceScene sc, rt_bvh_node* nodes, uint2* pairs, uint32_t padded,
                                                       rt_bvh_node* s_nodes, uint2* s_pairs, uint32_t s_padded, KeyBox box, uint32_t* clear, uint32_t* s_clear)
{
    uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf < sc.n_tris) { if (clear) clear[leaf] = 0u; if (s_clear) s_clear[leaf] = 0u; }
    if (leaf >= sc.n_tris) {
        if (nodes && leaf < padded) pairs[leaf] = make_uint2(UINT_MAX, UINT_MAX);
        if (leaf < s_padded) s_pairs[leaf] = make_uint2(UINT_MAX, UINT_MAX);
        return;
    }
    float4 a = sc.v0[leaf], b = sc.v1[leaf], c = sc.v2[leaf];
    float lo[3] = { RMIN(RMIN(a.x, b.x), c.x), RMIN(RMIN(a.y, b.y), c.y), RMIN(RMIN(a.z, b.z), c.z) };
    float hi[3] = { RMAX(RMAX(a.x, b.x), c.x), RMAX(RMAX(a.y, b.y), c.y), RMAX(RMAX(a.z, b.z), c.z) };
    for (int k = 0; k < 3; k++) {
        uint32_t flat = (hi[k] - lo[k]) < FLT_EPSILON;
        hi[k] += (float)flat * FLT_EPSILON;
        lo[k] -= (float)flat * FLT_EPSILON;
    }
    float cx = (a.x + b.x + c.x) / 3.0f;
    float cy = (a.y + b.y + c.y) / 3.0f;
    float cz = (a.z + b.z + c.z) / 3.0f;
    float nx = normalize_scale(sc.bb_lo[0], sc.bb_hi[0], cx);
    float ny = normalize_scale(sc.bb_lo[1], sc.bb_hi[1], cy);
    float nz = normalize_scale(sc.bb_lo[2], sc.bb_hi[2], cz);
    uint4 ref = sc.ref[leaf];
    rt_bvh_node nd;
    nd.data[0] = 0;            // parent: written by the radix-tree kernel
    nd.data[1] = ref.x;        // triangleIndex
    nd.data[2] = ref.y;        // submeshIndex
    nd.data[3] = ref.z;        // modelIndex
    nd.boundsMin[0] = lo[0]; nd.boundsMin[1] = lo[1]; nd.boundsMin[2] = lo[2]; nd.boundsMin[3] = 1.0f; // LEAF_NODE
    nd.boundsMax[0] = hi[0]; nd.boundsMax[1] = hi[1]; nd.boundsMax[2] = hi[2]; nd.boundsMax[3] = 0.0f;
    if (nodes) {
        nodes[leaf] = nd;
        pairs[leaf] = make_uint2(morton3d(nx, ny, nz), leaf);
    }
    if (s_nodes) {
        s_nodes[leaf] = nd;
        s_pairs[leaf] = make_uint2(search_tree_key(lo, hi, box), leaf);
    }
}

void launch_morton_leaves(hipStream_t s, const DeviceScene& sc, rt_bvh_node* nodes, uint2* pairs, uint32_t padded,
                          rt_bvh_node* s_nodes, uint2* s_pairs, uint32_t s_padded, uint32_t* clear, uint32_t* s_clear, bool adaptive_keys)
{
    uint32_t threads = nodes && padded > sc.n_tris ? padded : sc.n_tris;            // (nodes == nullptr: the search tree's outputs only)
    if (s_nodes && s_padded > threads) threads = s_padded;
    KeyBox box;
    for (int k = 0; k < 3; k++) { box.lo[k] = sc.bb_lo[k]; box.hi[k] = sc.bb_hi[k]; }
    key_layout(box);
    if (!adaptive_keys) box.seq = 0ull;
    hipLaunchKernelGGL(k_morton_leaves, dim3((threads + 255) / 256), dim3(256), 0, s, sc, nodes, pairs, padded, s_nodes, s_pairs, s_nodes ? s_padded : 0u, box, clear, s_clear);
}

// ----------------------------------------------------------------------------------------
// 3. radix tree over the sorted (morton, leaf) array: what determineRange / findSplit / constructNode compute
//    (BVH.h:138-321, the duplicate-key variant of Karras 2012), expressed through two monotone predicates.
//
//    Sorted keys share fewer and fewer leading bits with a fixed key the further away they lie, so both searches of the
//    reference are searches for the LAST position at which a monotone predicate still holds, and any bisection finds
//    the same position as the reference's doubling / halving loops:
//      span of inner node i   : i itself plus the longest stretch, towards the neighbour that shares MORE leading bits with
//                               key[i], of keys sharing more leading bits with key[i] than the other neighbour does;
//                               inside a run of equal keys (both neighbours equal) the node covers the rest of the run;
//                               node 0 covers everything;
//      split of a span [f, l] : the last position whose key shares more leading bits with key[f] than key[l] does
//                               (f itself when key[f] == key[l]).
//    Children: positions [f, split] and [split + 1, l]; a one-element side is the leaf stored at that position.
//    `built` = number of inner nodes constructed: n - 1 (FIXED), or n - 2 as the reference kernel's guard has it
//    (idx < leafCount - 2, BVHKernels.cl:58: inner node n - 2 is never built; LITERAL_ZERO).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int shared_top_bits(uint32_t a, uint32_t b) { return a == b ? 32 : __clz((int)(a ^ b)); }

struct KeySpan { uint32_t first, last; };

__device__ KeySpan inner_node_span(const uint2* __restrict__ keys, uint32_t i, uint32_t n)
{
    if (i == 0u) return KeySpan{ 0u, n - 1u };
    const uint32_t mine = keys[i].x, before = keys[i - 1u].x, after = keys[i + 1u].x;
    if (before == mine && after == mine) {
        // the rest of the run of equal keys: the keys are sorted, so "keys[i + steps] == mine" holds up to the run's end and never after --
        // doubling, then bisection (a step-by-step scan made this kernel quadratic in the run length: 684 us per tree on the Sponza-class
        // stand-in, whose longest run has ~10 800 keys, 63 % of its construct())
        const uint32_t room = n - 1u - i;
        auto same = [&](uint32_t steps) { return steps <= room && keys[i + steps].x == mine; };
        uint32_t yes = 1u, no = 2u;                                  // same(1) holds: after == mine
        while (same(no)) { yes = no; no <<= 1; }
        while (no - yes > 1u) {
            const uint32_t mid = yes + ((no - yes) >> 1);
            if (same(mid)) yes = mid; else no = mid;
        }
        return KeySpan{ i, i + yes };
    }
    const int with_before = shared_top_bits(mine, before), with_after = shared_top_bits(mine, after);
    const bool backwards = with_before > with_after;
    const int floor_bits = backwards ? with_after : with_before;     // what the far neighbour shares: the span must beat it
    const uint32_t room = backwards ? i : n - 1u - i;                // steps that stay inside the array
    auto belongs = [&](uint32_t steps) {
        return steps <= room && shared_top_bits(mine, keys[backwards ? i - steps : i + steps].x) > floor_bits;
    };
    uint32_t yes = 0u, no = 1u;                                      // belongs(0) holds trivially
    while (belongs(no)) { yes = no; no <<= 1; }
    while (no - yes > 1u) {
        const uint32_t mid = yes + ((no - yes) >> 1);
        if (belongs(mid)) yes = mid; else no = mid;
    }
    return backwards ? KeySpan{ i - yes, i } : KeySpan{ i, i + yes };
}

__device__ uint32_t span_split(const uint2* __restrict__ keys, KeySpan sp)
{
    const uint32_t head = keys[sp.first].x, tail = keys[sp.last].x;
    if (head == tail) return sp.first;
    const int ends_share = shared_top_bits(head, tail);
    uint32_t yes = sp.first, no = sp.last;                            // the head shares everything with itself, the tail too little
    while (no - yes > 1u) {
        const uint32_t mid = yes + ((no - yes) >> 1);
        if (shared_top_bits(head, keys[mid].x) > ends_share) yes = mid; else no = mid;
    }
    return yes;
}

// (clear2: two words the refit behind this kernel accumulates into -- tree depth, NaN flag -- cleared here instead of by two fill launches)
// (second: the search tree's buffers, built by the same launch -- blockIdx.y = 1; same n, same number of nodes)
struct RadixTreeSet { rt_bvh_node* nodes; const uint2* keys; uint2* spans; uint32_t* clear2; };
__global__ __launch_bounds__(256) void k_radix_tree(RadixTreeSet first, RadixTreeSet second, uint32_t n, uint32_t built)
{
    const RadixTreeSet set = blockIdx.y ? second : first;
    rt_bvh_node* const nodes = set.nodes;
    const uint2* const __restrict__ keys = set.keys;
    uint2* const __restrict__ spans = set.spans;
    uint32_t* const clear2 = set.clear2;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u && clear2 != nullptr) { clear2[0] = 0u; clear2[1] = 0u; }
    if (i >= built) return;
    const KeySpan sp = inner_node_span(keys, i, n);
    const uint32_t split = span_split(keys, sp);
    const uint32_t self = n + i;
    if (spans) spans[i] = make_uint2(sp.first, sp.last);      // for the range-query refit
    // a side of one position is that position's leaf, otherwise the inner node numbered like the side's boundary position
    const uint32_t child_a = sp.first == split ? keys[split].y : n + split;
    const uint32_t child_b = sp.last == split + 1u ? keys[split + 1u].y : n + split + 1u;
    rt_bvh_node* nd = &nodes[self];
    // init box (FLT_MAX, FLT_MIN) -- FLT_MIN is the smallest POSITIVE normal, as written (BVH.h:309-310)
    nd->boundsMin[0] = FLT_MAX; nd->boundsMin[1] = FLT_MAX; nd->boundsMin[2] = FLT_MAX; nd->boundsMin[3] = 0.0f; // INNER_NODE
    nd->boundsMax[0] = FLT_MIN; nd->boundsMax[1] = FLT_MIN; nd->boundsMax[2] = FLT_MIN; nd->boundsMax[3] = 0.0f;
    if (i == 0u) nd->data[0] = UINT_MAX;                               // the root has no parent (BVH.h:312-313)
    nd->data[1] = child_a;
    nd->data[2] = child_b;
    nd->data[3] = 0;
    nodes[child_a].data[0] = self;
    nodes[child_b].data[0] = self;
}

void launch_radix_tree(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, uint32_t n, uint32_t built, uint2* spans, uint32_t* clear2,
                       rt_bvh_node* nodes2, const uint2* pairs2, uint2* spans2, uint32_t* clear2_2)
{
    if (built == 0u) return;
    const RadixTreeSet first{ nodes, pairs, spans, clear2 }, second{ nodes2, pairs2, spans2, clear2_2 };
    hipLaunchKernelGGL(k_radix_tree, dim3((built + 255) / 256, nodes2 ? 2u : 1u), dim3(256), 0, s, first, second, n, built);
}

// ----------------------------------------------------------------------------------------
// 4. refit: computeBoundingBoxes, BVHKernels.cl:65-79 + mergeBoundingBox BVH.h:328-334 +
//    merge3 AABB.h:107-113 (all four lanes).  One walker per leaf; the SECOND arriver at a
//    node merges (decided by the atomic's return value; the reference re-reads the counter
//    non-atomically and has no fence).  Boxes written by one CU are handed to another CU
//    with an agent-scope release before the counter add and an agent-scope acquire after it
//    (per-CU L1 and per-XCD L2 are not coherent, MI355X_MICROARCH "inter-workgroup visibility").
// ----------------------------------------------------------------------------------------
// mergeBoundingBox (BVH.h:328-334) = merge3 (AABB.h:107-113) on all four lanes: child A, child B, then the node's own box,
// which still holds what constructNode gave it: (FLT_MAX, FLT_MAX, FLT_MAX, 0) / (FLT_MIN, FLT_MIN, FLT_MIN, 0).  The
// operand order matters for ties (RMIN / RMAX return their second operand unless the first is strictly smaller / larger).
// TIGHT (the search tree, whose boxes are ours to choose as long as they nest): the union and nothing else.  The
// reference's init box makes every maximum at least FLT_MIN, i.e. every box of a model's negative half reach to the plane 0 -- on
// ManFigure (x in [-8, 8], z in [-2.2, 1.2]) that trebles the summed surface area of the inner boxes, and with it the boxes a ray
// enters for nothing (round 3: 47 -> 13 four-wide steps per 8x8 block, tools/tree_quality.py).
template <bool TIGHT>
__device__ __forceinline__ void refit_merge(float4 alo, float4 ahi, float4 blo, float4 bhi, float4& rlo, float4& rhi)
{
    if (TIGHT) {
        rlo.x = RMIN(alo.x, blo.x); rlo.y = RMIN(alo.y, blo.y); rlo.z = RMIN(alo.z, blo.z); rlo.w = 0.0f;
        rhi.x = RMAX(ahi.x, bhi.x); rhi.y = RMAX(ahi.y, bhi.y); rhi.z = RMAX(ahi.z, bhi.z); rhi.w = 0.0f;
        return;
    }
    const float4 slo = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.0f), shi = make_float4(FLT_MIN, FLT_MIN, FLT_MIN, 0.0f);
    float4 t;
    t.x = RMIN(alo.x, blo.x); t.y = RMIN(alo.y, blo.y); t.z = RMIN(alo.z, blo.z); t.w = RMIN(alo.w, blo.w);
    rlo.x = RMIN(t.x, slo.x); rlo.y = RMIN(t.y, slo.y); rlo.z = RMIN(t.z, slo.z); rlo.w = RMIN(t.w, slo.w);
    t.x = RMAX(ahi.x, bhi.x); t.y = RMAX(ahi.y, bhi.y); t.z = RMAX(ahi.z, bhi.z); t.w = RMAX(ahi.w, bhi.w);
    rhi.x = RMAX(t.x, shi.x); rhi.y = RMAX(t.y, shi.y); rhi.z = RMAX(t.z, shi.z); rhi.w = RMAX(t.w, shi.w);
}

#define REFIT_RUN 8
// One 16-byte access at device scope (sc1: performed at the level every XCD's L2 and every CU's L1 agree on).  The hand-shake at a node
// with two inner children needs the first arriver's box to be visible to the second -- and nothing else: an agent-scope release /
// acquire FENCE pair gives that by writing back and invalidating the caches of the whole CU / XCD for every walker that passes such a
// node, which is what a refit cost (ClumsyDragon, 50 032 leaves: 230 us for either tree, whatever its depth -- 58 % of construct()).
// With the boxes that another walker reads written by sc1 stores, the walker's wait for its own stores (vmcnt) before the counter's
// atomic, and the second arriver's sc1 loads behind the atomic's return, no cache-wide operation is left (the same protocol as the
// sub-tree sharing of k_trace_packet4, trace_kernels.hip).
typedef float v4f_dev __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(float* p, float4 v)
{
    v4f_dev x = { v.x, v.y, v.z, v.w };
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 load16_sc1(const float* p)
{
    v4f_dev x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
    return make_float4(x.x, x.y, x.z, x.w);
}
// One walker per pair of sibling leaves climbs towards the root and carries the box it has just computed in registers.
//   * a node whose other child is a LEAF is merged on the spot: a leaf's box is final, no hand-shake is needed -- this is
//     every step along a chain of duplicate Morton codes (node i = leaf + node i+1), the deep part of the reference tree;
//   * a node whose other child is an INNER node is merged by the second arriver, decided by the counter's return value
//     (the reference re-reads the counter non-atomically and has no fence); the first arriver leaves its box in memory
//     behind an agent-scope release, the second reads it after an agent-scope acquire (per-CU L1 and per-XCD L2 are not
//     coherent, MI355X_MICROARCH "inter-workgroup visibility").
// Same merges in the same operand order as one walker per leaf with a counter at every node, hence the same bytes.
// only_if != null: the launch does nothing unless *only_if is non-zero (the climb as the fall-back of the range-query refit, decided on the
// device: no read-back in the middle of a build)
// (k_refit carries the climbs of both hierarchies of a build: blockIdx.y = 1 walks the second set)
struct RefitSet { rt_bvh_node* nodes; uint32_t* counters; uint32_t* depth_out; const uint32_t* only_if; uint32_t tight; };
template <bool TIGHT>
__device__ __forceinline__ void refit_climb(rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t* depth_out, const uint32_t* only_if)
{
    if (only_if != nullptr && *only_if == 0u) return;
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    uint32_t current = nodes[leaf].data[0];                // the leaf's parent
    if (current == UINT_MAX) return;
    uint32_t ca = nodes[current].data[1], cb = nodes[current].data[2];
    if (ca != leaf || cb >= n) return;                     // walkers start at (leaf, leaf) nodes, from child A
    float4 lo, hi;
    refit_merge<TIGHT>(*reinterpret_cast<const float4*>(nodes[ca].boundsMin), *reinterpret_cast<const float4*>(nodes[ca].boundsMax),
                *reinterpret_cast<const float4*>(nodes[cb].boundsMin), *reinterpret_cast<const float4*>(nodes[cb].boundsMax), lo, hi);
    // The walker also carries the height of its sub-tree (a leaf has height 0): the root's height is the tree depth the
    // traversal stacks are sized from, and no separate pass has to climb from every leaf to the root for it.  At a node
    // with two inner children the first arriver leaves its height in the upper bits of the arrival counter.
    uint32_t height = 1;
    for (;;) {
        store16_sc1(nodes[current].boundsMin, lo);          // (the sibling's walker may be the one that reads it: see store16_sc1)
        store16_sc1(nodes[current].boundsMax, hi);
        // Along a chain of duplicate Morton codes node i is (leaf, node i + 1): the nodes above `current` are current - 1,
        // current - 2, ...  Eight levels are taken per round trip: their records and their leaves' boxes are fetched together,
        // the chain shape is checked, and the eight merges run in registers in the same order as one by one.
        if (current >= n + REFIT_RUN) {
            uint4 rec[REFIT_RUN];
#pragma unroll
            for (int k = 0; k < REFIT_RUN; k++) rec[k] = *reinterpret_cast<const uint4*>(nodes[current - 1u - (uint32_t)k].data);
            bool chain = nodes[current].data[0] == current - 1u;
#pragma unroll
            for (int k = 0; k < REFIT_RUN; k++)
                chain = chain && rec[k].y < n && rec[k].z == current - (uint32_t)k &&          // (leaf, the node below)
                        (k + 1 == REFIT_RUN || rec[k].x == current - 2u - (uint32_t)k);          // its parent is the next one up
            if (chain) {
                float4 llo[REFIT_RUN], lhi[REFIT_RUN];
#pragma unroll
                for (int k = 0; k < REFIT_RUN; k++) {
                    llo[k] = *reinterpret_cast<const float4*>(nodes[rec[k].y].boundsMin);
                    lhi[k] = *reinterpret_cast<const float4*>(nodes[rec[k].y].boundsMax);
                }
#pragma unroll
                for (int k = 0; k < REFIT_RUN; k++) {
                    float4 rlo, rhi;
                    refit_merge<TIGHT>(llo[k], lhi[k], lo, hi, rlo, rhi);                   // child A = the leaf, child B = the chain below
                    lo = rlo; hi = rhi;
                    if (k + 1 < REFIT_RUN) {
                        *reinterpret_cast<float4*>(nodes[current - 1u - (uint32_t)k].boundsMin) = lo;
                        *reinterpret_cast<float4*>(nodes[current - 1u - (uint32_t)k].boundsMax) = hi;
                    }
                }
                current -= REFIT_RUN;
                height += REFIT_RUN;
                continue;                                                            // the loop head stores the last one
            }
        }
        const uint32_t parent = nodes[current].data[0];
        if (parent == UINT_MAX) { *depth_out = height; break; }            // the root: only one walker gets here
        ca = nodes[parent].data[1]; cb = nodes[parent].data[2];
        const bool i_am_a = ca == current;
        const uint32_t sibling = i_am_a ? cb : ca;
        uint32_t sibling_height = 0;
        float4 slo, shi;
        if (sibling >= n) {
            // inner sibling: whoever comes second merges.  My box stands in memory (sc1 stores) once my stores are acknowledged ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t old = __hip_atomic_fetch_add(&counters[parent - n], 1u + (height << 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // two bits of arrivals, 30 of height
            if (old == 0) break;                           // first arriver leaves; its box is in memory, its height in the counter
            sibling_height = old >> 2;
            // ... and the first arriver's was there before its atomic was: read where it was written
            slo = load16_sc1(nodes[sibling].boundsMin);
            shi = load16_sc1(nodes[sibling].boundsMax);
        } else {
            slo = *reinterpret_cast<const float4*>(nodes[sibling].boundsMin);       // a leaf's box: written by an earlier kernel
            shi = *reinterpret_cast<const float4*>(nodes[sibling].boundsMax);
        }
        height = (height > sibling_height ? height : sibling_height) + 1u;
        float4 rlo, rhi;
        if (i_am_a) refit_merge<TIGHT>(lo, hi, slo, shi, rlo, rhi);
        else refit_merge<TIGHT>(slo, shi, lo, hi, rlo, rhi);
        lo = rlo; hi = rhi;
        current = parent;
    }
}

__global__ __launch_bounds__(256) void k_refit(RefitSet first, RefitSet second, uint32_t n)
{
    const RefitSet a = blockIdx.y ? second : first;
    if (a.tight) refit_climb<true>(a.nodes, a.counters, n, a.depth_out, a.only_if);
    else refit_climb<false>(a.nodes, a.counters, n, a.depth_out, a.only_if);
}

void launch_refit(hipStream_t s, rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t* depth_out, bool tight, const uint32_t* only_if)
{
    const RefitSet first{ nodes, counters, depth_out, only_if, tight ? 1u : 0u };
    hipLaunchKernelGGL(k_refit, dim3((n + 255) / 256, 1), dim3(256), 0, s, first, first, n);
}
// the reference tree (its own merge) and the search tree (plain unions) in one launch
void launch_refit_pair(hipStream_t s, uint32_t n, rt_bvh_node* nodes, uint32_t* counters, uint32_t* depth_out, const uint32_t* only_if,
                       rt_bvh_node* s_nodes, uint32_t* s_counters, uint32_t* s_depth_out, const uint32_t* s_only_if, bool s_tight)
{
    const RefitSet first{ nodes, counters, depth_out, only_if, 0u }, second{ s_nodes, s_counters, s_depth_out, s_only_if, s_tight ? 1u : 0u };
    hipLaunchKernelGGL(k_refit, dim3((n + 255) / 256, 2), dim3(256), 0, s, first, second, n);
}

// ----------------------------------------------------------------------------------------
// 4a. refit by RANGE QUERIES (large scenes).  The climb above is serial in the depth of the reference tree, and duplicate
//     Morton codes make that tree a chain (10 835 levels on the Sponza-class stand-in, 129 512 on a 1 M-triangle height
//     field).  But every inner node covers a contiguous span [first, last] of the sorted leaves, children split the span in
//     order (A = [first, split], B = [split + 1, last]), and the reference's merges are
//         lo = RMIN(RMIN(A, B), FLT_MAX),  hi = RMAX(RMAX(A, B), FLT_MIN),   RMIN(a, b) = a < b ? a : b  (the RIGHT operand wins ties).
//     On NaN-free boxes "right wins ties" makes RMIN the RIGHTMOST minimum of an ordered sequence -- associative, and
//     idempotent over overlapping pieces as long as pieces are combined left to right (+0 / -0 included: they compare equal,
//     the rightmost is kept) -- and the clamps against the init box commute with it.  So a node's box is the rightmost
//     minimum / maximum over its span of the leaf boxes in sorted order, clamped once: the same bits as the climb, without
//     the climb.  Query structure: per 64-leaf block, running folds from the block's start (prefix) and to its end (suffix);
//     a sparse table over whole blocks; a span inside one block is folded directly.  NaN boxes (not associative) are
//     detected and send the build back to the climb.  The tree depth (what the climb carried as the root's height) is the
//     largest number of spans covering a leaf position: +1 / -1 marks, the library's scan, a maximum.
// ----------------------------------------------------------------------------------------
#define RQ_BLOCK 64u
struct BoxPair { float4 lo, hi; };
__device__ __forceinline__ BoxPair rm_fold(const BoxPair& a, const BoxPair& b)       // a before b in leaf order
{
    BoxPair r;
    r.lo.x = RMIN(a.lo.x, b.lo.x); r.lo.y = RMIN(a.lo.y, b.lo.y); r.lo.z = RMIN(a.lo.z, b.lo.z); r.lo.w = 0.0f;
    r.hi.x = RMAX(a.hi.x, b.hi.x); r.hi.y = RMAX(a.hi.y, b.hi.y); r.hi.z = RMAX(a.hi.z, b.hi.z); r.hi.w = 0.0f;
    return r;
}

// one thread per 64-leaf block: leaf boxes in sorted order, the folds from the block's start and to its end, the block's fold
__global__ __launch_bounds__(64) void k_rq_blocks(const rt_bvh_node* __restrict__ nodes, const uint2* __restrict__ pairs, uint32_t n,
                                                  float4* __restrict__ slo, float4* __restrict__ shi, float4* __restrict__ plo, float4* __restrict__ phi,
                                                  float4* __restrict__ qlo, float4* __restrict__ qhi, float4* __restrict__ tlo, float4* __restrict__ thi,
                                                  uint32_t* __restrict__ nan_flag)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t first = b * RQ_BLOCK;
    if (first >= n) return;
    const uint32_t end = first + RQ_BLOCK < n ? first + RQ_BLOCK : n;
    BoxPair acc{};
    bool nan = false;
    for (uint32_t j = first; j < end; j++) {
        const rt_bvh_node* lf = &nodes[pairs[j].y];
        BoxPair me;
        me.lo = make_float4(lf->boundsMin[0], lf->boundsMin[1], lf->boundsMin[2], 0.0f);
        me.hi = make_float4(lf->boundsMax[0], lf->boundsMax[1], lf->boundsMax[2], 0.0f);
        nan = nan || me.lo.x != me.lo.x || me.lo.y != me.lo.y || me.lo.z != me.lo.z || me.hi.x != me.hi.x || me.hi.y != me.hi.y || me.hi.z != me.hi.z;
        slo[j] = me.lo; shi[j] = me.hi;
        acc = j == first ? me : rm_fold(acc, me);
        plo[j] = acc.lo; phi[j] = acc.hi;
    }
    tlo[b] = acc.lo; thi[b] = acc.hi;
    for (uint32_t j = end; j-- > first;) {
        BoxPair me; me.lo = slo[j]; me.hi = shi[j];
        acc = j + 1 == end ? me : rm_fold(me, acc);
        qlo[j] = acc.lo; qhi[j] = acc.hi;
    }
    if (nan) atomicOr(nan_flag, 1u);
}

// sparse table over blocks, one level per launch: out[b] = fold(in[b], in[b + half]) for b + 2 half <= blocks
__global__ __launch_bounds__(256) void k_rq_level(const float4* __restrict__ in_lo, const float4* __restrict__ in_hi, float4* __restrict__ out_lo,
                                                  float4* __restrict__ out_hi, uint32_t blocks, uint32_t half)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b + 2u * half > blocks) return;
    BoxPair a, c;
    a.lo = in_lo[b]; a.hi = in_hi[b]; c.lo = in_lo[b + half]; c.hi = in_hi[b + half];
    const BoxPair r = rm_fold(a, c);
    out_lo[b] = r.lo; out_hi[b] = r.hi;
}

__global__ __launch_bounds__(256) void k_rq_refit(rt_bvh_node* __restrict__ nodes, const uint2* __restrict__ spans, uint32_t n,
                                                  const float4* __restrict__ slo, const float4* __restrict__ shi, const float4* __restrict__ plo,
                                                  const float4* __restrict__ phi, const float4* __restrict__ qlo, const float4* __restrict__ qhi,
                                                  const float4* __restrict__ tlo, const float4* __restrict__ thi, uint32_t blocks, int32_t* __restrict__ marks)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1u) return;
    const uint2 sp = spans[i];
    const uint32_t bf = sp.x / RQ_BLOCK, bl = sp.y / RQ_BLOCK;
    BoxPair acc;
    if (bf == bl) {
        acc.lo = slo[sp.x]; acc.hi = shi[sp.x];
        for (uint32_t j = sp.x + 1u; j <= sp.y; j++) { BoxPair me; me.lo = slo[j]; me.hi = shi[j]; acc = rm_fold(acc, me); }
    } else {
        acc.lo = qlo[sp.x]; acc.hi = qhi[sp.x];
        if (bl > bf + 1u) {
            const uint32_t cnt = bl - bf - 1u;
            const uint32_t k = 31u - (uint32_t)__clz((int)cnt);                 // largest power of two <= cnt
            const float4* Llo = tlo + (size_t)k * blocks; const float4* Lhi = thi + (size_t)k * blocks;
            BoxPair a, c;
            a.lo = Llo[bf + 1u]; a.hi = Lhi[bf + 1u];
            c.lo = Llo[bl - (1u << k)]; c.hi = Lhi[bl - (1u << k)];
            acc = rm_fold(acc, rm_fold(a, c));
        }
        BoxPair tail; tail.lo = plo[sp.y]; tail.hi = phi[sp.y];
        acc = rm_fold(acc, tail);
    }
    // the clamps of merge3 against the node's init box (FLT_MAX, FLT_MIN): value preserving, applied once
    rt_bvh_node* nd = &nodes[n + i];
    nd->boundsMin[0] = RMIN(acc.lo.x, FLT_MAX); nd->boundsMin[1] = RMIN(acc.lo.y, FLT_MAX); nd->boundsMin[2] = RMIN(acc.lo.z, FLT_MAX); nd->boundsMin[3] = 0.0f;
    nd->boundsMax[0] = RMAX(acc.hi.x, FLT_MIN); nd->boundsMax[1] = RMAX(acc.hi.y, FLT_MIN); nd->boundsMax[2] = RMAX(acc.hi.z, FLT_MIN); nd->boundsMax[3] = 0.0f;
    atomicAdd(&marks[sp.x], 1);
    atomicAdd(&marks[sp.y + 1u], -1);
}

__global__ __launch_bounds__(256) void k_rq_depth(const uint32_t* __restrict__ covered, uint32_t n, uint32_t* __restrict__ depth_out)
{
    uint32_t m = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) m = covered[j] > m ? covered[j] : m;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_down((int)m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63u) == 0u) atomicMax(depth_out, m);
}

size_t refit_rq_scratch_bytes(uint32_t n)
{
    const size_t blocks = (n + RQ_BLOCK - 1) / RQ_BLOCK;
    size_t levels = 1; while ((1ull << levels) <= blocks) levels++;
    return (size_t)n * 6 * sizeof(float4) + blocks * levels * 2 * sizeof(float4) + ((size_t)n + 1) * 2 * sizeof(uint32_t) + scan_scratch_words(n + 1) * sizeof(uint32_t) + 256;
}

// boxes of all inner nodes + the tree depth; *nan_flag_out (device word, zeroed here) != 0 afterwards: NaN boxes, use launch_refit
void launch_refit_rq(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, const uint2* spans, uint32_t n, void* scratch, uint32_t* depth_out,
                     uint32_t* nan_flag_out)
{
    const uint32_t blocks = (n + RQ_BLOCK - 1) / RQ_BLOCK;
    uint32_t levels = 1; while ((1ull << levels) <= blocks) levels++;
    float4* f = (float4*)scratch;
    float4 *slo = f, *shi = f + n, *plo = f + 2 * (size_t)n, *phi = f + 3 * (size_t)n, *qlo = f + 4 * (size_t)n, *qhi = f + 5 * (size_t)n;
    float4* tlo = f + 6 * (size_t)n; float4* thi = tlo + (size_t)blocks * levels;
    int32_t* marks = (int32_t*)(thi + (size_t)blocks * levels);
    uint32_t* covered = (uint32_t*)marks + (n + 1);
    uint32_t* scan_scratch = covered + (n + 1);
    (void)hipMemsetAsync(marks, 0, ((size_t)n + 1) * sizeof(int32_t), s);
    (void)hipMemsetAsync(nan_flag_out, 0, sizeof(uint32_t), s);
    (void)hipMemsetAsync(depth_out, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_rq_blocks, dim3((blocks + 63) / 64), dim3(64), 0, s, nodes, pairs, n, slo, shi, plo, phi, qlo, qhi, tlo, thi, nan_flag_out);
    for (uint32_t k = 1; k < levels; k++)
        hipLaunchKernelGGL(k_rq_level, dim3((blocks + 255) / 256), dim3(256), 0, s, tlo + (size_t)(k - 1) * blocks, thi + (size_t)(k - 1) * blocks,
                           tlo + (size_t)k * blocks, thi + (size_t)k * blocks, blocks, 1u << (k - 1));
    hipLaunchKernelGGL(k_rq_refit, dim3((n - 1 + 255) / 256), dim3(256), 0, s, nodes, spans, n, slo, shi, plo, phi, qlo, qhi, tlo, thi, blocks, marks);
    launch_scan_inclusive(s, (const uint32_t*)marks, covered, n, scan_scratch);         // two's complement: the running sums are the cover counts
    hipLaunchKernelGGL(k_rq_depth, dim3(256), dim3(256), 0, s, covered, n, depth_out);
}

// ----------------------------------------------------------------------------------------
// 4a'. refit by SPANS (round 4; every scene from 256 triangles on, both hierarchies).  The same fact as 4a -- a node's box is the ordered fold
//      of the leaf boxes over its contiguous span of the sorted leaves, clamped once -- with a query structure that costs O(n) to build and is
//      read without any walker waiting for another: a segment tree over the sorted leaf boxes (level k, entry m = the fold of positions
//      [m 2^k, (m + 1) 2^k) that exist; levels 0-8 inside a block's LDS, the few entries above by one block), and per inner node the
//      standard bottom-up decomposition of its span into O(log n) entries, left parts folded left to right, right parts right to left
//      (RMIN / RMAX keep the RIGHT operand on ties: the order is part of the bytes).  The climb (4.) hands every box through the device's
//      coherence point twice per level -- 65-80 us per tree at 50 032 leaves, serial in the depth; 4a's sparse table costs O(n log n) and
//      300 us at 262 656 leaves.  NaN boxes (not associative) are detected while the leaves are gathered and send the build to the climb.
// ----------------------------------------------------------------------------------------
#define SEG_MAX_LEVELS 30
struct SegLevels { uint32_t off[SEG_MAX_LEVELS]; uint32_t cnt[SEG_MAX_LEVELS]; uint32_t levels; };
static SegLevels seg_levels(uint32_t n)
{
    SegLevels L{};
    uint32_t cnt = n, off = 0, k = 0;
    for (;;) {
        L.off[k] = off; L.cnt[k] = cnt; k++;
        if (cnt <= 1u || k == SEG_MAX_LEVELS) break;
        off += cnt; cnt = (cnt + 1u) >> 1;
    }
    L.levels = k;
    return L;
}
static size_t seg_entries(uint32_t n) { const SegLevels L = seg_levels(n); return (size_t)L.off[L.levels - 1] + L.cnt[L.levels - 1]; }

// (the four kernels carry both hierarchies of a build: blockIdx.y = 1 works on the second set -- same n, same levels)
struct SegSet { rt_bvh_node* nodes; const uint2* pairs; const uint2* spans; float4* lo; float4* hi; int32_t* marks; int32_t* tiles; uint32_t* ticket;
                uint32_t* depth_out; uint32_t* nan_flag; uint32_t tight; };
__global__ __launch_bounds__(256) void k_seg_low(SegSet first, SegSet second, uint32_t n, SegLevels L)
{
    __shared__ float4 s_lo[256], s_hi[256];
    const SegSet set = blockIdx.y ? second : first;
    const rt_bvh_node* const __restrict__ nodes = set.nodes;
    const uint2* const __restrict__ pairs = set.pairs;
    float4* const __restrict__ lo = set.lo;
    float4* const __restrict__ hi = set.hi;
    uint32_t* const __restrict__ nan_flag = set.nan_flag;
    int32_t* const __restrict__ marks = set.marks;
    const uint32_t t = threadIdx.x, base = blockIdx.x * 256u, j = base + t;
    uint32_t live = n - base < 256u ? n - base : 256u;                   // entries of this block at the current level
    BoxPair me{};
    if (t < live) {
        const rt_bvh_node* lf = &nodes[pairs[j].y];
        me.lo = make_float4(lf->boundsMin[0], lf->boundsMin[1], lf->boundsMin[2], 0.0f);
        me.hi = make_float4(lf->boundsMax[0], lf->boundsMax[1], lf->boundsMax[2], 0.0f);
        if (me.lo.x != me.lo.x || me.lo.y != me.lo.y || me.lo.z != me.lo.z || me.hi.x != me.hi.x || me.hi.y != me.hi.y || me.hi.z != me.hi.z) atomicOr(nan_flag, 1u);
        lo[j] = me.lo; hi[j] = me.hi;                                     // level 0 (L.off[0] == 0)
        marks[j] = 0;                                                     // (the span marks of k_seg_refit start at zero: no fill launch)
        if (j + 1u == n) marks[n] = 0;
    }
    s_lo[t] = me.lo; s_hi[t] = me.hi;
    __syncthreads();
    for (uint32_t k = 1; k <= 8u && k < L.levels; k++) {
        const uint32_t below = live;
        live = (live + 1u) >> 1;
        BoxPair a{};
        if (t < live) {
            a.lo = s_lo[2u * t]; a.hi = s_hi[2u * t];
            if (2u * t + 1u < below) { BoxPair b; b.lo = s_lo[2u * t + 1u]; b.hi = s_hi[2u * t + 1u]; a = rm_fold(a, b); }
        }
        __syncthreads();
        if (t < live) {
            s_lo[t] = a.lo; s_hi[t] = a.hi;
            const size_t at = (size_t)L.off[k] + (base >> k) + t;
            lo[at] = a.lo; hi[at] = a.hi;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_seg_top(SegSet first, SegSet second, SegLevels L)
{
    float4* const __restrict__ lo = blockIdx.y ? second.lo : first.lo;
    float4* const __restrict__ hi = blockIdx.y ? second.hi : first.hi;
    for (uint32_t k = 9; k < L.levels; k++) {
        const float4* plo = lo + L.off[k - 1]; const float4* phi = hi + L.off[k - 1];
        for (uint32_t m = threadIdx.x; m < L.cnt[k]; m += 1024u) {
            BoxPair a; a.lo = plo[2u * m]; a.hi = phi[2u * m];
            if (2u * m + 1u < L.cnt[k - 1]) { BoxPair b; b.lo = plo[2u * m + 1u]; b.hi = phi[2u * m + 1u]; a = rm_fold(a, b); }
            lo[(size_t)L.off[k] + m] = a.lo; hi[(size_t)L.off[k] + m] = a.hi;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// marks[key] += sign, once per run of consecutive lanes with the same key: the nodes of a chain of equal Morton codes are consecutive and share
// an end of their spans, i.e. a wave would send up to 64 adds to one address (the active lanes of a wave are a prefix of it here)
__device__ __forceinline__ void seg_mark(int32_t* marks, uint32_t key, int32_t sign)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool head = lane == 0u || prev != key;
    const uint64_t active = __ballot(true), heads = __ballot(head);
    if (head) {
        const uint64_t above = (heads >> lane) >> 1;
        const uint32_t run_end = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 64u;
        const uint32_t live_end = (uint32_t)__popcll(active);
        atomicAdd(&marks[key], sign * (int32_t)((run_end < live_end ? run_end : live_end) - lane));
    }
}

__global__ __launch_bounds__(256) void k_seg_refit(SegSet first, SegSet second, uint32_t n, SegLevels L, uint32_t top_in_lds)
{
    const SegSet set = blockIdx.y ? second : first;
    rt_bvh_node* const __restrict__ nodes = set.nodes;
    const uint2* const __restrict__ spans = set.spans;
    const float4* const __restrict__ lo = set.lo;
    const float4* const __restrict__ hi = set.hi;
    int32_t* const __restrict__ marks = set.marks;
    const bool TIGHT = set.tight != 0u;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // Up to 65 536 leaves the levels above the blocks' own (level 8: one entry per 256 leaves, at most 256 of them) are folded by every block
    // for itself, in LDS, from the level-8 entries k_seg_low left -- the same folds in the same order as k_seg_top, which then is not launched
    __shared__ float4 t_lo[512], t_hi[512];                  // level k >= 8 at SEG_TOP_AT(k): 0, 256, 384, 448, ...
#define SEG_TOP_AT(k) (512u - (512u >> ((k) - 8u)))
    if (top_in_lds) {
        const uint32_t t = threadIdx.x;
        if (t < L.cnt[8]) { t_lo[t] = lo[(size_t)L.off[8] + t]; t_hi[t] = hi[(size_t)L.off[8] + t]; }
        __syncthreads();
        for (uint32_t k = 9; k < L.levels; k++) {
            if (t < L.cnt[k]) {
                const uint32_t below = SEG_TOP_AT(k - 1u);
                BoxPair x; x.lo = t_lo[below + 2u * t]; x.hi = t_hi[below + 2u * t];
                if (2u * t + 1u < L.cnt[k - 1]) { BoxPair y; y.lo = t_lo[below + 2u * t + 1u]; y.hi = t_hi[below + 2u * t + 1u]; x = rm_fold(x, y); }
                t_lo[SEG_TOP_AT(k) + t] = x.lo; t_hi[SEG_TOP_AT(k) + t] = x.hi;
            }
            __syncthreads();
        }
    }
    if (i >= n - 1u) return;
    const uint2 sp = spans[i];
    uint32_t a = sp.x, b = sp.y;
    BoxPair left{}, right{};
    bool have_left = false, have_right = false;
    // Which entries cover the span is arithmetic on (a, b) alone: six levels' worth of them are determined first, fetched together, and folded
    // in the order of the one-level-at-a-time walk (left parts left to right, right parts right to left) -- three round trips through the
    // cache hierarchy instead of one per level (50 032 leaves: 17 levels).
    bool done = false;
    for (uint32_t k0 = 0; k0 < L.levels && !done; k0 += 6u) {
        bool tl[6], tr[6];
        size_t il[6], ir[6];
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            tl[j] = false; tr[j] = false; il[j] = 0; ir[j] = 0;
            const uint32_t k = k0 + j;
            if (done || k >= L.levels) continue;
            const size_t at = (top_in_lds && k >= 8u) ? (size_t)SEG_TOP_AT(k) : (size_t)L.off[k];
            if (a & 1u) { tl[j] = true; il[j] = at + a; a++; }                        // a right child: nothing to its left belongs to the span
            if (!(b & 1u)) {                                                          // a left child: nothing to its right belongs to the span
                tr[j] = true; ir[j] = at + b;
                if (b == 0u) done = true;                                             // (a > b follows: a >= 1 here unless the span was [0, 0], which no inner node has)
                else b--;
            }
            if (!done) { if (a > b) done = true; else { a >>= 1; b >>= 1; } }
        }
        BoxPair el[6], er[6];
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            const bool lds = top_in_lds && k0 + j >= 8u;
            if (tl[j]) { if (lds) { el[j].lo = t_lo[il[j]]; el[j].hi = t_hi[il[j]]; } else { el[j].lo = lo[il[j]]; el[j].hi = hi[il[j]]; } }
            if (tr[j]) { if (lds) { er[j].lo = t_lo[ir[j]]; er[j].hi = t_hi[ir[j]]; } else { er[j].lo = lo[ir[j]]; er[j].hi = hi[ir[j]]; } }
        }
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            if (tl[j]) { left = have_left ? rm_fold(left, el[j]) : el[j]; have_left = true; }
            if (tr[j]) { right = have_right ? rm_fold(er[j], right) : er[j]; have_right = true; }
        }
    }
    const BoxPair acc = (have_left && have_right) ? rm_fold(left, right) : (have_left ? left : right);
    rt_bvh_node* nd = &nodes[n + i];
    if (TIGHT) {
        nd->boundsMin[0] = acc.lo.x; nd->boundsMin[1] = acc.lo.y; nd->boundsMin[2] = acc.lo.z; nd->boundsMin[3] = 0.0f;
        nd->boundsMax[0] = acc.hi.x; nd->boundsMax[1] = acc.hi.y; nd->boundsMax[2] = acc.hi.z; nd->boundsMax[3] = 0.0f;
    } else {
        // the clamps of merge3 against the node's init box (FLT_MAX, FLT_MIN): value preserving, applied once (as in 4a)
        nd->boundsMin[0] = RMIN(acc.lo.x, FLT_MAX); nd->boundsMin[1] = RMIN(acc.lo.y, FLT_MAX); nd->boundsMin[2] = RMIN(acc.lo.z, FLT_MAX); nd->boundsMin[3] = 0.0f;
        nd->boundsMax[0] = RMAX(acc.hi.x, FLT_MIN); nd->boundsMax[1] = RMAX(acc.hi.y, FLT_MIN); nd->boundsMax[2] = RMAX(acc.hi.z, FLT_MIN); nd->boundsMax[3] = 0.0f;
    }
    seg_mark(marks, sp.x, 1);
    seg_mark(marks, sp.y + 1u, -1);
}

// The tree depth = the largest number of spans covering a leaf position = the maximum of the running sums of the +1 / -1 marks: computed by the
// build's last kernel (k_finish, section 7a) -- every block takes a tile of 256 marks (total and maximum of the running sum), leaves the pair where
// the others can see it, and the block that finishes LAST (a ticket counter) combines the tiles in order.  (4a does this with the library's
// multi-launch scan and a maximum.)
__device__ __forceinline__ int32_t seg_block_exclusive(int32_t v, int32_t* sh /* 256 */, int32_t* total)
{
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const int32_t o = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += o;
        __syncthreads();
    }
    const int32_t inc = sh[t];
    *total = sh[255];
    __syncthreads();
    return inc - v;
}
size_t refit_span_scratch_bytes(uint32_t n)
{
    return seg_entries(n) * 2 * sizeof(float4) + ((size_t)n + 1) * 2 * sizeof(uint32_t) + scan_scratch_words(n + 1) * sizeof(uint32_t) + 256;
}

static SegSet seg_set(rt_bvh_node* nodes, const uint2* pairs, const uint2* spans, uint32_t n, void* scratch, uint32_t* depth_out, uint32_t* nan_flag_out, bool tight)
{
    const size_t entries = seg_entries(n);
    SegSet a;
    a.nodes = nodes; a.pairs = pairs; a.spans = spans;
    a.lo = (float4*)scratch; a.hi = a.lo + entries;
    a.marks = (int32_t*)(a.hi + entries);
    // (tiles: the depth tiles' (total, maximum) pairs; ticket: zero between builds -- cleared once when the scratch is allocated)
    a.tiles = a.marks + (n + 1);
    a.ticket = (uint32_t*)a.tiles + (n + 1);
    a.depth_out = depth_out; a.nan_flag = nan_flag_out; a.tight = tight ? 1u : 0u;
    return a;
}
// boxes of all inner nodes + the tree depth; *nan_flag_out (device word) != 0 afterwards: NaN boxes, the climb must redo it.
// (depth_out and nan_flag_out are consecutive words cleared by the k_radix_tree launch in front of this refit: launch_radix_tree's clear2.)
// nodes2 != null: the second hierarchy of the build (same n) in the same four launches.
// depth_later: the marks are left for launch_finish, which turns them into the depth on its way (one launch less).
void launch_refit_span(hipStream_t s, rt_bvh_node* nodes, const uint2* pairs, const uint2* spans, uint32_t n, void* scratch, uint32_t* depth_out,
                       uint32_t* nan_flag_out, bool tight, rt_bvh_node* nodes2, const uint2* pairs2, const uint2* spans2, void* scratch2,
                       uint32_t* depth_out2, uint32_t* nan_flag_out2, bool tight2, bool depth_later)
{
    const SegLevels L = seg_levels(n);
    const SegSet first = seg_set(nodes, pairs, spans, n, scratch, depth_out, nan_flag_out, tight);
    const SegSet second = nodes2 ? seg_set(nodes2, pairs2, spans2, n, scratch2, depth_out2, nan_flag_out2, tight2) : first;
    const uint32_t sets = nodes2 ? 2u : 1u;
    hipLaunchKernelGGL(k_seg_low, dim3((n + 255) / 256, sets), dim3(256), 0, s, first, second, n, L);
    const uint32_t top_in_lds = (L.levels > 9u && L.cnt[8] <= 256u) ? 1u : 0u;
    if (L.levels > 9u && !top_in_lds) hipLaunchKernelGGL(k_seg_top, dim3(1, sets), dim3(1024), 0, s, first, second, L);
    hipLaunchKernelGGL(k_seg_refit, dim3((n - 1 + 255) / 256, sets), dim3(256), 0, s, first, second, n, L, top_in_lds);
    (void)depth_later;
}

// ----------------------------------------------------------------------------------------
// 4b. LITERAL_ZERO refit (compatibility mode): computeBoundingBoxes as launched by
//     BVHManager.cpp:214-217 on zero-initialised buffers, made deterministic: one walker per work item of the
//     64-wide launch (items beyond the last leaf are clamped to it, BVHKernels.cl:67), every arrival counted, every
//     arrival after the first merges (the reference re-reads the counter: `> 1`), a parent link outside [n, 2n-1) ends the
//     walk (the children of the never-built node n-2 still hold the 0 the buffer was cleared to; the reference would
//     index its counters out of range there).  The genuine walkers run in parallel -- at most two arrive per node, the
//     second merges final children, decided by the atomic's return value -- and the clamped extra walkers, whose order
//     matters, run afterwards on one thread in work-item order, exactly as the oracle serialises them.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ void literal_merge(rt_bvh_node* nodes, uint32_t cur)
{
    const uint32_t a = nodes[cur].data[1], b = nodes[cur].data[2];
    // (the children's boxes may have been written by other walkers of this launch: device-scope loads, see store16_sc1)
    const float4 alo = load16_sc1(nodes[a].boundsMin), ahi = load16_sc1(nodes[a].boundsMax);
    const float4 blo = load16_sc1(nodes[b].boundsMin), bhi = load16_sc1(nodes[b].boundsMax);
    const float4 slo = load16_sc1(nodes[cur].boundsMin), shi = load16_sc1(nodes[cur].boundsMax);
    float4 t, lo, hi;                                  // merge3 (AABB.h:107-113) on all four lanes: (A, B), then the node's own box
    t.x = RMIN(alo.x, blo.x); t.y = RMIN(alo.y, blo.y); t.z = RMIN(alo.z, blo.z); t.w = RMIN(alo.w, blo.w);
    lo.x = RMIN(t.x, slo.x); lo.y = RMIN(t.y, slo.y); lo.z = RMIN(t.z, slo.z); lo.w = RMIN(t.w, slo.w);
    t.x = RMAX(ahi.x, bhi.x); t.y = RMAX(ahi.y, bhi.y); t.z = RMAX(ahi.z, bhi.z); t.w = RMAX(ahi.w, bhi.w);
    hi.x = RMAX(t.x, shi.x); hi.y = RMAX(t.y, shi.y); hi.z = RMAX(t.z, shi.z); hi.w = RMAX(t.w, shi.w);
    store16_sc1(nodes[cur].boundsMin, lo);
    store16_sc1(nodes[cur].boundsMax, hi);
}

__global__ __launch_bounds__(256) void k_refit_literal(rt_bvh_node* nodes, uint32_t* counters, uint32_t n)
{
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    uint32_t current = nodes[leaf].data[0];
    while (current != UINT_MAX) {
        if (current < n || current >= 2u * n - 1u) break;
        // As in k_refit: the box this walker merged last stands in memory (sc1 stores, acknowledged: vmcnt) before its arrival is counted,
        // and the second arriver reads the children's boxes with device-scope loads behind the atomic's return.  (Round 3 had agent-scope
        // fences here, whose cache write-back / invalidate completes asynchronously: once in some thousand builds a second arriver read a
        // child box of before the first arriver's merge -- three nodes with boxes too small.  The loads were plain loads then, which a
        // non-coherent L1 / L2 may serve; now no access of the hand-shake goes through a cache that another walker's write can miss.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&counters[current - n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) break;
        literal_merge(nodes, current);
        current = nodes[current].data[0];
    }
}

__global__ void k_refit_literal_extras(rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t extras, uint32_t* depth_out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (uint32_t e = 0; e < extras; e++) {
        uint32_t current = nodes[n - 1u].data[0];
        while (current != UINT_MAX) {
            if (current < n || current >= 2u * n - 1u) break;
            const uint32_t c = ++counters[current - n];
            if (c <= 1u) break;
            literal_merge(nodes, current);
            current = nodes[current].data[0];
        }
    }
    *depth_out = 0u;                                   // k_depth_literal takes the maximum
}

// longest chain of valid parent links above a leaf (the tree is broken on purpose in this mode: walks end at bad links)
__global__ __launch_bounds__(256) void k_depth_literal(const rt_bvh_node* nodes, uint32_t n, uint32_t* depth_out)
{
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    uint32_t d = 0, current = nodes[leaf].data[0];
    while (current != UINT_MAX && current >= n && current < 2u * n - 1u && d < 2u * n) { d++; current = nodes[current].data[0]; }
    atomicMax(depth_out, d);
}

void launch_refit_literal(hipStream_t s, rt_bvh_node* nodes, uint32_t* counters, uint32_t n, uint32_t* depth_out)
{
    hipLaunchKernelGGL(k_refit_literal, dim3((n + 255) / 256), dim3(256), 0, s, nodes, counters, n);
    const uint32_t launched = ((n + 63u) / 64u) * 64u;              // closestMultipleTo(leafCount, warp), BVHManager.cpp:213
    hipLaunchKernelGGL(k_refit_literal_extras, dim3(1), dim3(64), 0, s, nodes, counters, n, launched - n, depth_out);
    hipLaunchKernelGGL(k_depth_literal, dim3((n + 255) / 256), dim3(256), 0, s, nodes, n, depth_out);
}

// ----------------------------------------------------------------------------------------
// 5. re-layout into 64-byte traversal records + tree depth (sizes the deep-stack spill area)
// ----------------------------------------------------------------------------------------
// (the kernel: k_finish, section 7a)

// ----------------------------------------------------------------------------------------
// 6. search tree: a second hierarchy over the SAME leaves (same 48-byte leaf nodes, same boxes), ordered by Morton codes
//    of the box centres properly normalised to the model's box (the reference's normalizeScale divides by the extent
//    without subtracting the minimum, RTKernelUtils.h:63-69, which is what makes its tree 94-10 835 levels deep).
//    Only the key generation is new: the tree itself is built, refitted and re-laid-out by the kernels above.
//    DESIGN.md section 2 shows why primary rays may walk this tree instead of the reference's and still produce the
//    reference's contacts bit for bit.
// ----------------------------------------------------------------------------------------
// (normalised to the MODEL's box of the scene header -- the box the reference's own codes are scaled by, RTKernelUtils.h:63-69, here with the
//  minimum subtracted -- passed by value: the keys need the leaves and nothing of the reference tree, so the two hierarchies can be built
//  side by side.  Through round 3 the reference ROOT's box was read, i.e. the reference refit had to finish first.)
// ----------------------------------------------------------------------------------------
// 7. four-wide records for the packet kernel: node i with its inner children replaced by their children.  Skipping the
//    inner children's own boxes is allowed for the same reason the search tree is (any nested hierarchy over the leaves
//    reaches the same leaves); one wave step then descends two levels.
// ----------------------------------------------------------------------------------------
// (the kernel: k_finish below)

// ----------------------------------------------------------------------------------------
// 7a. the end of a build in ONE launch (through round 3 and most of round 4 three: re-layout, four-wide records, tree depth), for both
//     hierarchies (blockIdx.y), plus the hand-over of the depth / flag words to the host's pinned copy.  A block of 256 threads serves inner
//     nodes / sorted positions / marks [256 b, 256 b + 256):
//       * traversal record j; rank and packed triangle record of the leaf at sorted position j -- no thread reads what another writes;
//       * four-wide record j (sets with quad != null): the slot with the largest surface area opened, twice (section 7), with the children's
//         records assembled from the node buffer the way record j is, instead of read back from `trav` (which other blocks of this launch are
//         still writing); a thread assembles one 128-byte record, the block hands its 256 records over through LDS so that the eight 16-byte
//         parts of a record are written by eight neighbouring lanes (whole lines per store instruction instead of 64 lines touched by each of
//         eight: 36 -> 7 us at 100 062 records);
//       * sets refitted by spans (marks != null): the block's 256 marks as one tile of the depth computation -- total and maximum of the running
//         sum, left where the others see them; the block that finishes LAST (a ticket) combines the tiles in order.
// ----------------------------------------------------------------------------------------
struct FinishSet {
    const rt_bvh_node* nodes; const uint2* pairs; TravNode* trav; uint32_t* pos; uint32_t id_offset;
    QuadNode* quad;                                          // record j of this set at quad[j]; null: no four-wide records
    const int32_t* marks; int32_t* tiles; uint32_t* ticket;  // marks == null: words[0] holds the depth already (a climb wrote it)
    uint32_t* words;                                         // device: [0] depth, [1] NaN flag of the span refit
    uint32_t* host_words;                                    // where the host reads them (pinned); null: a copy follows
};
__device__ __forceinline__ TravNode make_trav(const rt_bvh_node* nodes, uint32_t n, uint32_t id_offset, uint32_t j)
{
    const rt_bvh_node* nd = &nodes[n + j];
    const uint32_t ca = nd->data[1], cb = nd->data[2];
    TravNode t;
    for (int k = 0; k < 3; k++) {
        t.a_lo[k] = nodes[ca].boundsMin[k]; t.a_hi[k] = nodes[ca].boundsMax[k];
        t.b_lo[k] = nodes[cb].boundsMin[k]; t.b_hi[k] = nodes[cb].boundsMax[k];
    }
    t.child_a = ca < n ? (RT_LEAF_FLAG | ca) : ca - n + id_offset;
    t.child_b = cb < n ? (RT_LEAF_FLAG | cb) : cb - n + id_offset;
    t.pad0 = 0; t.pad1 = 0;
    return t;
}
__global__ __launch_bounds__(256) void k_finish(FinishSet set0, FinishSet set1, uint32_t n, DeviceScene sc, TriRec* __restrict__ trirec_first, uint32_t greedy)
{
    __shared__ float4 stage[256 * 9];
    __shared__ int32_t sh[256];
    __shared__ int32_t sh_max[256];
    __shared__ uint32_t s_last;
    const FinishSet set = blockIdx.y ? set1 : set0;
    const rt_bvh_node* const nodes = set.nodes;
    const uint32_t t = threadIdx.x, j = blockIdx.x * 256u + t;
    TravNode rec{};
    if (j < n - 1u) { rec = make_trav(nodes, n, set.id_offset, j); set.trav[j] = rec; }
    if (j < n) {
        const uint32_t leaf = set.pairs[j].y;
        set.pos[leaf] = j;
        if (blockIdx.y == 0u && trirec_first != nullptr) {
            const float4 v0 = sc.v0[leaf], e1 = sc.e1[leaf], e2 = sc.e2[leaf];
            TriRec r;
            r.v0[0] = v0.x; r.v0[1] = v0.y; r.v0[2] = v0.z;
            r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z;
            r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
            r.pos = j; r.material = __float_as_uint(v0.w); r.pad = 0u;
            trirec_first[leaf] = r;
        }
    }
    if (set.quad != nullptr) {                                            // (uniform per block)
        float lo[4][3], hi[4][3];
        uint32_t id[4];
        int m = 0;
        if (j < n - 1u) {
            if (greedy) {
                for (int k = 0; k < 3; k++) { lo[0][k] = rec.a_lo[k]; hi[0][k] = rec.a_hi[k]; lo[1][k] = rec.b_lo[k]; hi[1][k] = rec.b_hi[k]; }
                id[0] = rec.child_a; id[1] = rec.child_b; m = 2;
                for (int round = 0; round < 2; round++) {
                    int best = -1; float best_area = -1.0f;
                    for (int c = 0; c < m; c++) {
                        if (id[c] & RT_LEAF_FLAG) continue;
                        const float dx = hi[c][0] - lo[c][0], dy = hi[c][1] - lo[c][1], dz = hi[c][2] - lo[c][2];
                        const float area = dx * dy + dy * dz + dz * dx;
                        if (!(area <= best_area)) { best = c; best_area = area; }       // (a NaN area is opened like any other)
                    }
                    if (best < 0) break;
                    const TravNode u = make_trav(nodes, n, set.id_offset, id[best] - set.id_offset);
                    for (int k = 0; k < 3; k++) { lo[best][k] = u.a_lo[k]; hi[best][k] = u.a_hi[k]; lo[m][k] = u.b_lo[k]; hi[m][k] = u.b_hi[k]; }
                    id[best] = u.child_a; id[m] = u.child_b; m++;
                }
            } else {
                for (int c = 0; c < 2; c++) {
                    const uint32_t cid = c ? rec.child_b : rec.child_a;
                    const float* clo = c ? rec.b_lo : rec.a_lo;
                    const float* chi = c ? rec.b_hi : rec.a_hi;
                    if (cid & RT_LEAF_FLAG) {
                        for (int k = 0; k < 3; k++) { lo[m][k] = clo[k]; hi[m][k] = chi[k]; }
                        id[m++] = cid;
                    } else {
                        const TravNode u = make_trav(nodes, n, set.id_offset, cid - set.id_offset);
                        for (int k = 0; k < 3; k++) { lo[m][k] = u.a_lo[k]; hi[m][k] = u.a_hi[k]; }
                        id[m++] = u.child_a;
                        for (int k = 0; k < 3; k++) { lo[m][k] = u.b_lo[k]; hi[m][k] = u.b_hi[k]; }
                        id[m++] = u.child_b;
                    }
                }
            }
        }
        // an unused slot: id = RT_SENTINEL and a box of NaN planes, which no ray enters under any of the kernels' tests (every compare with
        // a NaN is false; the camera-relative copies get an empty box of their own, k_prepare_camera)
        for (; m < 4; m++) { for (int k = 0; k < 3; k++) { lo[m][k] = __builtin_nanf(""); hi[m][k] = __builtin_nanf(""); } id[m] = RT_SENTINEL; }
        float4* mine = stage + t * 9u;
        mine[0] = make_float4(lo[0][0], lo[0][1], lo[0][2], __uint_as_float(id[0]));
        mine[1] = make_float4(hi[0][0], hi[0][1], hi[0][2], __uint_as_float(id[1]));
        mine[2] = make_float4(lo[1][0], lo[1][1], lo[1][2], __uint_as_float(id[2]));
        mine[3] = make_float4(hi[1][0], hi[1][1], hi[1][2], __uint_as_float(id[3]));
        mine[4] = make_float4(lo[2][0], lo[2][1], lo[2][2], 0.0f);
        mine[5] = make_float4(hi[2][0], hi[2][1], hi[2][2], 0.0f);
        mine[6] = make_float4(lo[3][0], lo[3][1], lo[3][2], 0.0f);
        mine[7] = make_float4(hi[3][0], hi[3][1], hi[3][2], 0.0f);
        __syncthreads();
        const uint32_t first = blockIdx.x * 256u;
        const uint32_t live = first < n - 1u ? (n - 1u - first < 256u ? n - 1u - first : 256u) : 0u;
        float4* out = reinterpret_cast<float4*>(set.quad + first);
        for (uint32_t p = t; p < live * 8u; p += 256u) out[p] = stage[(p >> 3) * 9u + (p & 7u)];
    }
    if (blockIdx.x == 0u && t == 0u && set.host_words != nullptr) set.host_words[1] = set.words[1];
    if (set.marks == nullptr) {
        if (blockIdx.x == 0u && t == 0u && set.host_words != nullptr) set.host_words[0] = set.words[0];
        return;
    }
    // the depth: this block's tile of the marks, one mark per thread
    const uint32_t tiles = gridDim.x;
    const int32_t v = j < n ? set.marks[j] : 0;
    int32_t total;
    const int32_t ex = seg_block_exclusive(v, sh, &total);
    sh_max[t] = ex + v;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (t < o) sh_max[t] = sh_max[t] > sh_max[t + o] ? sh_max[t] : sh_max[t + o]; __syncthreads(); }
    if (t == 0) {
        const int32_t mx = sh_max[0] > 0 ? sh_max[0] : 0;
        __hip_atomic_store(&set.tiles[2u * blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&set.tiles[2u * blockIdx.x + 1u], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // my tile stands in memory before my ticket is drawn
        s_last = __hip_atomic_fetch_add(set.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tiles - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    int32_t carry = 0, deepest = 0;
    for (uint32_t base = 0; base < tiles; base += 256u) {
        const uint32_t k = base + t;
        const int32_t tot = k < tiles ? __hip_atomic_load(&set.tiles[2u * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int32_t mx = k < tiles ? __hip_atomic_load(&set.tiles[2u * k + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int32_t round_total;
        const int32_t exc = seg_block_exclusive(tot, sh, &round_total);
        sh_max[t] = k < tiles ? carry + exc + mx : 0;
        __syncthreads();
        for (uint32_t o = 128; o > 0; o >>= 1) { if (t < o) sh_max[t] = sh_max[t] > sh_max[t + o] ? sh_max[t] : sh_max[t + o]; __syncthreads(); }
        deepest = sh_max[0] > deepest ? sh_max[0] : deepest;
        carry += round_total;
        __syncthreads();
    }
    if (t == 0) {
        set.words[0] = (uint32_t)deepest;
        if (set.host_words != nullptr) set.host_words[0] = (uint32_t)deepest;
        *set.ticket = 0u;                                                     // (left at zero for the next build)
    }
}

// span_scratch / span_scratch2: the scratch the span refit of that hierarchy ran on with depth_later (null: its depth word is final already)
void launch_finish(hipStream_t s, uint32_t n, bool quad_by_area, const DeviceScene& sc, TriRec* trirec,
                   const rt_bvh_node* nodes, const uint2* pairs, TravNode* trav, uint32_t* pos, uint32_t id_offset, QuadNode* quad, void* span_scratch,
                   uint32_t* words, uint32_t* host_words,
                   const rt_bvh_node* nodes2, const uint2* pairs2, TravNode* trav2, uint32_t* pos2, uint32_t id_offset2, QuadNode* quad2, void* span_scratch2,
                   uint32_t* words2, uint32_t* host_words2)
{
    const uint32_t greedy = quad_by_area ? 1u : 0u;             // (knob RTMI_QUAD_BY_AREA=0: both children opened instead of the largest slot, twice)
    auto make = [n](const rt_bvh_node* nd, const uint2* pr, TravNode* tv, uint32_t* ps, uint32_t off, QuadNode* q, void* scratch, uint32_t* w, uint32_t* hw) {
        FinishSet f{};
        f.nodes = nd; f.pairs = pr; f.trav = tv; f.pos = ps; f.id_offset = off; f.quad = q; f.words = w; f.host_words = hw;
        if (scratch) { const SegSet g = seg_set(nullptr, nullptr, nullptr, n, scratch, nullptr, nullptr, false); f.marks = g.marks; f.tiles = g.tiles; f.ticket = g.ticket; }
        return f;
    };
    const FinishSet first = make(nodes, pairs, trav, pos, id_offset, quad, span_scratch, words, host_words);
    const FinishSet second = nodes2 ? make(nodes2, pairs2, trav2, pos2, id_offset2, quad2, span_scratch2, words2, host_words2) : first;
    hipLaunchKernelGGL(k_finish, dim3((n + 255) / 256, nodes2 ? 2u : 1u), dim3(256), 0, s, first, second, n, sc, trirec, greedy);
}

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_build()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_gather_triangles)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}

// ----------------------------------------------------------------------------------------
// 9. the planes of the leaf boxes, sorted, for the host (rtmi_capi.hip, ensure_host_planes): per axis the 2n corner coordinates as
//    sortable keys (a float's bits, sign handled; NaN and the padding sort last), ordered by the library's bitonic network -- the
//    host copies the result back instead of sorting 2n floats three times itself (20 ms at 70 000 leaves, 80 ms at 260 000)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_keys(const rt_bvh_node* __restrict__ nodes, uint32_t n, uint32_t axis, uint2* __restrict__ pairs, uint32_t n_pow2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    uint32_t key = 0xFFFFFFFFu;
    if (i < 2u * n) {
        const float v = (i & 1u) ? nodes[i >> 1].boundsMax[axis] : nodes[i >> 1].boundsMin[axis];
        if (v == v) { const uint32_t b = __float_as_uint(v); key = (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
    }
    pairs[i] = make_uint2(key, i);
}
void launch_plane_keys(hipStream_t s, const rt_bvh_node* leaves, uint32_t n, uint32_t axis, uint2* pairs, uint32_t n_pow2)
{
    hipLaunchKernelGGL(k_plane_keys, dim3((n_pow2 + 255) / 256), dim3(256), 0, s, leaves, n, axis, pairs, n_pow2);
    launch_bitonic_sort(s, pairs, n_pow2);
}
