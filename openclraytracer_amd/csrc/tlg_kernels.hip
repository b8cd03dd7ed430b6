// tlg_kernels.hip -- Two-Level Grid (Kalojanov et al.) build and traversal for gfx950.
//
// Restates, with the FIXED readings of SURVEY.md 8(c) (see include/rtmi.h, rt_tlg_*):
//   build      OpenCLRayTracer/src/CLRayTracerEngine/TwoLevelGridKernels.cl:43-221 and
//              Include/CLData/AccelerationStructs/TwoLevelGrid.h:55-403, AABB.h:219-226,389-540
//   traversal  TwoLevelGrid.h:419-678 (tlg_generate_contact / processTopLevelCell), kernels TwoLevelGridKernels.cl:228-270
// Data the reference re-reads from the packed scene buffer per pair / per ray comes from the SoA triangle table.
#include "rtmi_internal.h"
#include "device_math.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "tlg_math.h"
#include "shade_math.h"

#define TLG_BLOCK 256

// ------------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------------

// float cell span of a triangle's bounding box: countOverlappingCells (TwoLevelGrid.h:86-99) and
// writeOverlappingPairs (:146-150) evaluate the same expressions
__device__ __forceinline__ void top_span(const rt_grid_data& g, float4 a, float4 b, float4 c, float s[3], float e[3])
{
    const float lo[3] = { RMIN(a.x, RMIN(b.x, c.x)), RMIN(a.y, RMIN(b.y, c.y)), RMIN(a.z, RMIN(b.z, c.z)) };
    const float hi[3] = { RMAX(a.x, RMAX(b.x, c.x)), RMAX(a.y, RMAX(b.y, c.y)), RMAX(a.z, RMAX(b.z, c.z)) };
    const float org[3] = { g.box_lo[0], g.box_lo[1], g.box_lo[2] };
    const float step[3] = { g.stepX, g.stepY, g.stepZ };
    const float last[3] = { (float)(g.resX - 1), (float)(g.resY - 1), (float)(g.resZ - 1) };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float fs = floorf((lo[k] - org[k]) / step[k]);
        float fe = floorf((hi[k] - org[k]) / step[k]);
        s[k] = RMIN(fs, last[k]);
        e[k] = RMIN(fe, last[k]);
    }
}

// prepareDataKernel -> prepareGridData (TwoLevelGridKernels.cl:43-51; all n triangles)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_count_top(rt_grid_data g, const float4* __restrict__ v0, const float4* __restrict__ v1,
                                                             const float4* __restrict__ v2, uint32_t n, uint32_t* __restrict__ counters)
{
    uint32_t i = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (i >= n) return;
    float s[3], e[3];
    top_span(g, v0[i], v1[i], v2[i], s, e);
    float cx = (e[0] - s[0]) + 1.0f, cy = (e[1] - s[1]) + 1.0f, cz = (e[2] - s[2]) + 1.0f;
    counters[i] = tlg_f2u(cx * cy * cz);
}

// writePairsKernel -> writePairs -> writeOverlappingPairs (TwoLevelGrid.h:138-188)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_emit_top(rt_grid_data g, const float4* __restrict__ v0, const float4* __restrict__ v1,
                                                            const float4* __restrict__ v2, uint32_t n, const uint32_t* __restrict__ scan,
                                                            const uint32_t* __restrict__ counters, uint2* __restrict__ pairs, uint32_t pairs_cap)
{
    uint32_t i = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (i >= n) return;
    float s[3], e[3];
    top_span(g, v0[i], v1[i], v2[i], s, e);
    const uint32_t mx[3] = { g.resX - 1, g.resY - 1, g.resZ - 1 };
    uint32_t c0[3], c1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c0[k] = tlg_f2u(s[k]); c0[k] = c0[k] > mx[k] ? mx[k] : c0[k];
        c1[k] = tlg_f2u(e[k]); c1[k] = c1[k] > mx[k] ? mx[k] : c1[k];
    }
    const uint32_t cnt = counters[i];
    const uint32_t first = scan[i] - cnt;
    uint32_t w = 0;
    for (uint32_t z = c0[2]; z <= c1[2]; z++)
        for (uint32_t y = c0[1]; y <= c1[1]; y++)
            for (uint32_t x = c0[0]; x <= c1[0]; x++) {
                // (pairs_cap: a rebuild that runs on the previous build's array sizes never writes past them; its sizes are checked afterwards)
                if (w < cnt && first + w < pairs_cap) pairs[first + w] = make_uint2(z * g.resX * g.resY + y * g.resX + x, i);   // getCellIndex :55-59
                w++;
            }
}

// extractCellRangesKernel / extractLeafCellsKernel (TwoLevelGridKernels.cl:71-116,176-221): [first, end) of every key
// in the sorted array.  One thread per element over the whole array (the reference's LDS indexing only works in
// work-group 0), the last range closes at `count`, UINT_MAX holes never become a cell.
// `count_dev` (may be null): the element count lives on the device (a rebuild without host round trips); the grid then covers
// the array's capacity.  Keys at or beyond `ranges_cap` are never written (a stale capacity is detected by the host afterwards).
__global__ __launch_bounds__(TLG_BLOCK) void k_segment_bounds(const uint2* __restrict__ sorted, uint32_t count, const uint32_t* __restrict__ count_dev,
                                                              uint2* __restrict__ ranges, uint32_t ranges_cap)
{
    if (count_dev) { const uint32_t c = *count_dev; count = c < count ? c : count; }     // never beyond the array (`count` = its capacity then)
    uint32_t i = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t me = sorted[i].x;
    if (i + 1 < count) {
        const uint32_t right = sorted[i + 1].x;
        if (right > me) {
            if (me < ranges_cap) ranges[me].y = i + 1;
            if (right != 0xFFFFFFFFu && right < ranges_cap) ranges[right].x = i + 1;
        }
    } else if (me != 0xFFFFFFFFu && me < ranges_cap) ranges[me].y = count;
}

// countLeavesAndFillCellKernel -> fillTopLevelCell -> calcLeafCellResolution (TwoLevelGrid.h:193-225)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_cell_res(rt_grid_data g, const uint2* __restrict__ ranges, uint32_t cells,
                                                            rt_top_cell* __restrict__ top, uint32_t* __restrict__ leaves)
{
    uint32_t c = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (c >= cells) return;
    const uint2 r = ranges[c];
    const uint32_t prims = r.y - r.x;
    const float volume = g.stepX * g.stepY * g.stepZ;
    const float a = tlg_pow_third(g.leafDensity * (float)prims / volume);
    rt_top_cell t;
    t.resX = tlg_f2u(floorf(g.stepX * a)); t.resY = tlg_f2u(floorf(g.stepY * a)); t.resZ = tlg_f2u(floorf(g.stepZ * a));
    t.firstLeafIdx = 0;
    top[c] = t;
    leaves[c] = t.resX * t.resY * t.resZ;
}

// updateTopLevelCellsWithLeafRange (TwoLevelGridKernels.cl:129-136)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_first_leaf(rt_top_cell* __restrict__ top, const uint32_t* __restrict__ scan, uint32_t cells)
{
    uint32_t c = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (c == 0 || c >= cells) return;
    top[c].firstLeafIdx = scan[c - 1];
}

struct LeafGeom { float stepX, stepY, stepZ, baseX, baseY, baseZ; };
// leaf-cell geometry of one top-level cell (TwoLevelGrid.h:268-275 and :347-353: same expressions)
__device__ __forceinline__ LeafGeom leaf_geometry(const rt_grid_data& g, const rt_top_cell& tc, uint32_t topIdx)
{
    LeafGeom L;
    L.stepX = g.stepX / (float)tc.resX;
    L.stepY = g.stepY / (float)tc.resY;
    L.stepZ = g.stepZ / (float)tc.resZ;
    const uint32_t a = g.resX * g.resY;                           // getCellRefFromIndex :69-75
    const uint32_t z = topIdx / a, b = topIdx - a * z;
    const uint32_t cx = b % g.resX, cy = b / g.resX;
    L.baseX = g.box_lo[0] + (float)cx * g.stepX;
    L.baseY = g.box_lo[1] + (float)cy * g.stepY;
    L.baseZ = g.box_lo[2] + (float)z * g.stepZ;
    return L;
}

// calculateTriangleAABB, AABB.h:61-79 (one axis)
__device__ __forceinline__ void tri_axis_bounds(float a, float b, float c, float& lo, float& hi)
{
    lo = RMIN(a, b); hi = RMAX(a, b);
    lo = RMIN(lo, c); hi = RMAX(hi, c);
    const float flat = (float)(uint32_t)((hi - lo) < FLT_EPSILON);
    hi += flat * FLT_EPSILON;
    lo -= flat * FLT_EPSILON;
}

// one axis of AABBOverlaps (AABB.h:219-226) between leaf cell i and the triangle's box.  The 3-D predicate is the AND
// of three per-axis predicates that depend on that axis' index only, so the number of overlapping leaf cells of
// countOverlappingLeafCells' triple loop (TwoLevelGrid.h:277-291) is the product of three per-axis counts.
__device__ __forceinline__ bool axis_overlaps(float base, float step, uint32_t i, float tlo, float thi)
{
    const float lo = base + (float)i * step, hi = lo + step;
    return !(RMIN(hi, thi) < RMAX(lo, tlo));
}

__device__ __forceinline__ void axis_range(float base, float step, uint32_t res, float tlo, float thi,
                                           uint32_t& count, uint32_t& first, uint32_t& last)
{
    count = 0; first = 0; last = 0;
    for (uint32_t i = 0; i < res; i++)
        if (axis_overlaps(base, step, i, tlo, thi)) { if (!count) first = i; last = i; count++; }
}

// prepareGridDataForLeaves -> countLeafPairs -> countOverlappingLeafCells (TwoLevelGrid.h:262-323)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_count_leaf(rt_grid_data g, const uint2* __restrict__ pairs, uint32_t pairs_count,
                                                              const uint32_t* __restrict__ count_dev, uint32_t pairs_cap,
                                                              const rt_top_cell* __restrict__ top, const float4* __restrict__ v0,
                                                              const float4* __restrict__ v1, const float4* __restrict__ v2,
                                                              uint32_t* __restrict__ counters)
{
    if (count_dev) { const uint32_t c = *count_dev; pairs_count = c < pairs_cap ? c : pairs_cap; }
    uint32_t p = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (p >= pairs_count) {
        if (count_dev && p < pairs_cap) counters[p] = 0u;      // as prepareGridDataForLeaves does beyond the pairs (TwoLevelGridKernels.cl:154-155): the scan runs over the capacity
        return;
    }
    const uint2 pr = pairs[p];
    const rt_top_cell tc = top[pr.x];
    const float4 a = v0[pr.y], b = v1[pr.y], c = v2[pr.y];
    const LeafGeom L = leaf_geometry(g, tc, pr.x);
    float lo, hi;
    uint32_t nx, ny, nz, f, l;
    tri_axis_bounds(a.x, b.x, c.x, lo, hi); axis_range(L.baseX, L.stepX, tc.resX, lo, hi, nx, f, l);
    tri_axis_bounds(a.y, b.y, c.y, lo, hi); axis_range(L.baseY, L.stepY, tc.resY, lo, hi, ny, f, l);
    tri_axis_bounds(a.z, b.z, c.z, lo, hi); axis_range(L.baseZ, L.stepZ, tc.resZ, lo, hi, nz, f, l);
    counters[p] = (tc.resX && tc.resY && tc.resZ) ? nx * ny * nz : 0u;
}

// sign() of OpenCL C 6.12.4
__device__ __forceinline__ float cl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }

// AABBTriangleIntersect (AABB.h:477-540) with planeBoxOverlap (:455-475) and the AXISTEST macros (:389-453)
__device__ __forceinline__ bool aabb_triangle_intersect(f3 c, f3 h, f3 v0, f3 v1, f3 v2)
{
    bool yes = true;
    v0 = f3_sub(c, v0); v1 = f3_sub(c, v1); v2 = f3_sub(c, v2);
#define AXISTEST(pa_, pb_, rad_)                                            \
    do { const float pa = (pa_), pb = (pb_); const float mn = RMIN(pa, pb), mx = RMAX(pa, pb); const float rad = (rad_); \
         yes = yes && (mn <= rad && mx >= -rad); } while (0)
    {
        f3 e = f3_sub(v1, v0);
        float fex = fabsf(e.x), fey = fabsf(e.y), fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v2.y - e.y * v2.z, fez * h.y + fey * h.z);
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v2.x + e.x * v2.z, fez * h.x + fex * h.z);
        AXISTEST(e.y * v1.x - e.x * v1.y, e.y * v2.x - e.x * v2.y, fey * h.x + fex * h.y);
        e = f3_sub(v2, v1);
        fex = fabsf(e.x); fey = fabsf(e.y); fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v2.y - e.y * v2.z, fez * h.y + fey * h.z);
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v2.x + e.x * v2.z, fez * h.x + fex * h.z);
        AXISTEST(e.y * v0.x - e.x * v0.y, e.y * v1.x - e.x * v1.y, fey * h.x + fex * h.y);
        e = f3_sub(v0, v2);
        fex = fabsf(e.x); fey = fabsf(e.y); fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v1.y - e.y * v1.z, fez * h.y + fey * h.z);
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v1.x + e.x * v1.z, fez * h.x + fex * h.z);
        AXISTEST(e.y * v1.x - e.x * v1.y, e.y * v2.x - e.x * v2.y, fey * h.x + fex * h.y);
    }
#undef AXISTEST
    float mn, mx;
    mn = RMIN(v0.x, RMIN(v1.x, v2.x)); mx = RMAX(v0.x, RMAX(v1.x, v2.x));
    yes = yes && (mn <= h.x && mx >= -h.x);
    mn = RMIN(v0.y, RMIN(v1.y, v2.y)); mx = RMAX(v0.y, RMAX(v1.y, v2.y));
    yes = yes && (mn <= h.y && mx >= -h.y);
    mn = RMIN(v0.z, RMIN(v1.z, v2.z)); mx = RMAX(v0.z, RMAX(v1.z, v2.z));
    yes = yes && (mn <= h.z && mx >= -h.z);
    const f3 normal = f3_cross(f3_sub(v1, v0), f3_sub(v2, v1));
    f3 vmin, vmax;
    vmin.x = -cl_sign(normal.x) * h.x - v0.x; vmax.x = cl_sign(normal.x) * h.x - v0.x;
    vmin.y = -cl_sign(normal.y) * h.y - v0.y; vmax.y = cl_sign(normal.y) * h.y - v0.y;
    vmin.z = -cl_sign(normal.z) * h.z - v0.z; vmax.z = cl_sign(normal.z) * h.z - v0.z;
    yes = yes && (f3_dot(normal, vmin) <= 0.0f);
    yes = yes && (f3_dot(normal, vmax) >= 0.0f);
    return yes;
}

// writeLeafPairsKernel -> writeLeafPairs -> writeOverlappingLeafPairs (TwoLevelGrid.h:338-403).
// The reference runs the exact test on every leaf cell of the top-level cell; a cell can only pass when its box
// overlaps the triangle's box on every axis up to rounding, so the loops here cover the per-axis overlap ranges widened
// by one cell: the same cells pass, in the same z-y-x order.  At most `counted` pairs are written (FIXED reading: the
// reference would run into its neighbour's slots).
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_emit_leaf(rt_grid_data g, const uint2* __restrict__ pairs, uint32_t pairs_count,
                                                             const rt_top_cell* __restrict__ top, const float4* __restrict__ v0,
                                                             const float4* __restrict__ v1, const float4* __restrict__ v2,
                                                             const uint32_t* __restrict__ scan, const uint32_t* __restrict__ counters,
                                                             uint2* __restrict__ leaf_pairs, uint32_t* __restrict__ written_total,
                                                             const uint32_t* __restrict__ count_dev, uint32_t leaf_cap)
{
    if (count_dev) { const uint32_t c = *count_dev; pairs_count = c < pairs_count ? c : pairs_count; }   // `pairs_count` = the array's capacity then
    uint32_t p = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (p >= pairs_count) return;
    const uint32_t cnt = counters[p];
    if (cnt == 0) return;
    const uint2 pr = pairs[p];
    const rt_top_cell tc = top[pr.x];
    const float4 a4 = v0[pr.y], b4 = v1[pr.y], c4 = v2[pr.y];
    const LeafGeom L = leaf_geometry(g, tc, pr.x);
    float lo, hi;
    uint32_t n, fx, lx, fy, ly, fz, lz;
    tri_axis_bounds(a4.x, b4.x, c4.x, lo, hi); axis_range(L.baseX, L.stepX, tc.resX, lo, hi, n, fx, lx);
    tri_axis_bounds(a4.y, b4.y, c4.y, lo, hi); axis_range(L.baseY, L.stepY, tc.resY, lo, hi, n, fy, ly);
    tri_axis_bounds(a4.z, b4.z, c4.z, lo, hi); axis_range(L.baseZ, L.stepZ, tc.resZ, lo, hi, n, fz, lz);
    fx = fx ? fx - 1 : 0; fy = fy ? fy - 1 : 0; fz = fz ? fz - 1 : 0;
    lx = lx + 1 < tc.resX ? lx + 1 : tc.resX - 1; ly = ly + 1 < tc.resY ? ly + 1 : tc.resY - 1; lz = lz + 1 < tc.resZ ? lz + 1 : tc.resZ - 1;
    const f3 half = f3{ L.stepX * 0.5f, L.stepY * 0.5f, L.stepZ * 0.5f };
    const f3 A = f3_from4(a4), B = f3_from4(b4), C = f3_from4(c4);
    const uint32_t first = scan[p] - cnt;
    uint32_t w = 0;
    for (uint32_t z = fz; z <= lz; z++)
        for (uint32_t y = fy; y <= ly; y++)
            for (uint32_t x = fx; x <= lx; x++) {
                const f3 c = f3{ L.baseX + (float)x * L.stepX + half.x, L.baseY + (float)y * L.stepY + half.y,
                                 L.baseZ + (float)z * L.stepZ + half.z };
                if (aabb_triangle_intersect(c, half, A, B, C)) {
                    if (w < cnt && first + w < leaf_cap)
                        leaf_pairs[first + w] = make_uint2(z * tc.resX * tc.resY + y * tc.resX + x + tc.firstLeafIdx, pr.y);
                    w++;
                }
            }
    if (w > cnt) w = cnt;
    if (w) atomicAdd(written_total, w);
}

// ------------------------------------------------------------------------------------------------
// traversal
// ------------------------------------------------------------------------------------------------
struct TlgTraceArgs {
    rt_grid_data grid;
    const rt_top_cell* __restrict__ top;
    const uint2* __restrict__ leaf_ranges;
    const uint2* __restrict__ leaf_pairs;
    const float4* __restrict__ v0;
    const float4* __restrict__ e1;
    const float4* __restrict__ e2;
    const float4* __restrict__ nrm;
    const TriCam* __restrict__ tricam;      // primary rays: camera-relative triangle records (null: v0 / e1 / e2)
    const TlgTopRec* __restrict__ toprec;   // the walk's own records of the top-level cells and planes (null: the walk of round 4)
    const float4* __restrict__ axis_tab;    // resX + resY + resZ entries { lo, hi, RN(1 / (hi - lo)), 0 }
    const uint4* __restrict__ leafrec;      // per leaf cell, 32 bytes: { entries, triangle of entry 0, of entry 1, first entry in leaf_pairs } { triangles of entries 2 .. 5 }
    uint32_t cam_off;                       // primary rays: the planes relative to the camera position (TlgCamPlane) at this byte offset of toprec's buffer (0: none)
    uint32_t total_work;
    rt_camera cam;
    uint32_t rank, nranks, tiles_x, compact;
    const rt_ray* __restrict__ rays;
    rt_contact* __restrict__ out;
    float4* __restrict__ fused_rgba;        // primary, untiled: the pixel's colour as well (rt_tlg_generate_frame; null: contacts only)
    ShadeScene fused_scene;
};

// ---- the walk's own records (round 5) ---------------------------------------------------------------------------------------------
// A ray of config 4 walks through ~150 non-empty top-level cells and sets a leaf-level DDA up in each: six IEEE divisions per set-up,
// x * res / (hi - lo) and (tmax - tmin) / res per axis, whose quotients fix the cells visited and must be the reference's bit for bit.
// The divisors are no property of the ray: `res` belongs to the cell and (hi - lo) to the plane index.  With r = RN(1 / y) at hand a
// correctly rounded x / y takes five plain operations (Markstein: q0 = x r; e0 = fma(-y, q0, x); q1 = fma(e0, r, q0) is within an ulp,
// e1 = fma(-y, q1, x) is exact, q2 = fma(e1, r, q1) rounds correctly -- tools/micro/div_check.hip compares it with the IEEE division over
// every x for res = 1..2048 and over 2^36 (x, y) pairs) instead of the compiler's twelve (two v_div_scale, v_rcp, five FMAs, a multiply,
// v_div_fmas, v_div_fixup: 18.4 against 7.7 ns per wave-instruction-slot, tools/micro/pk_rate.hip).  It holds for operands away from
// overflow, underflow, zero divisors and non-numbers: a wave takes this walk only when all of its rays have |d| > 2^-40 on every axis
// and moderate origins, and a grid only when its box and steps are moderate (host check) -- everything else walks as in round 4.  (Bounds: |o|, |lo|, |hi| <= 2^30, |d| and hi - lo in [2^-20, 2^20], res <= 1023.)
// The records: per top-level cell { (float)res x3, firstLeafIdx, RN(1 / (float)res) x3, res packed 10:10:10 (0: empty cell) } and per
// plane index of an axis { lo, hi, RN(1 / (hi - lo)), hi - lo } exactly as the walk computes lo and hi (glo + (float)i * step, lo + step).
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_trace_records(rt_grid_data g, const rt_top_cell* __restrict__ top, uint32_t cells,
                                                                 TlgTopRec* __restrict__ rec, float4* __restrict__ axis_tab, uint32_t* __restrict__ too_fine)
{
    const uint32_t c = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (c < g.resX + g.resY + g.resZ) {
        uint32_t i = c; float org = g.box_lo[0], step = g.stepX;
        if (i >= g.resX) { i -= g.resX; org = g.box_lo[1]; step = g.stepY; if (i >= g.resY) { i -= g.resY; org = g.box_lo[2]; step = g.stepZ; } }
        const float lo = org + (float)i * step, hi = lo + step, w = hi - lo;
        // (a divisor whose significand is all ones is the one case the proofs of the sequence exclude; a width that is no ordinary number
        //  anyway: the whole grid then walks as in round 4)
        if (!(w >= 0x1p-20f && w <= 0x1p20f) || (__float_as_uint(w) & 0x7FFFFFu) == 0x7FFFFFu || !(fabsf(lo) <= 0x1p30f) || !(fabsf(hi) <= 0x1p30f)) atomicOr(too_fine, 1u);
        axis_tab[c] = make_float4(lo, hi, 1.0f / w, w);
    }
    if (c >= cells) return;
    const rt_top_cell t = top[c];
    TlgTopRec r;
    const bool empty = t.resX == 0 || t.resY == 0 || t.resZ == 0;
    if (!empty && (t.resX > 1023u || t.resY > 1023u || t.resZ > 1023u)) atomicOr(too_fine, 1u);
    r.fres[0] = (float)t.resX; r.fres[1] = (float)t.resY; r.fres[2] = (float)t.resZ;
    r.first_leaf = t.firstLeafIdx;
    r.rcp[0] = 1.0f / r.fres[0]; r.rcp[1] = 1.0f / r.fres[1]; r.rcp[2] = 1.0f / r.fres[2];
    r.packed = empty ? 0u : (t.resX | (t.resY << 10) | (t.resZ << 20));
    rec[c] = r;
}
// ... and per leaf cell { entries, the triangles of its first two entries, where its entries start } { the triangles of entries 2 .. 5 } (32 bytes):
// the walk meets a leaf cell's first triangles with ONE fetch instead of the range and then the pair behind it (two trips to the L2 in a row per
// leaf cell), and the packet walk, which holds the record in scalar registers, goes through up to six entries without a loop (99.9 % of the leaf
// cells a ray of config 4 meets hold at most six)
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_leaf_records(const uint2* __restrict__ leaf_ranges, const uint2* __restrict__ leaf_pairs, uint32_t leaf_cells,
                                                                uint32_t pairs_cap, uint4* __restrict__ rec)
{
    const uint32_t c = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (c >= leaf_cells) return;
    const uint2 r = leaf_ranges[c];
    uint4 o = make_uint4(0u, 0u, 0u, r.x), o2 = make_uint4(0u, 0u, 0u, 0u);
    if (r.x < r.y && r.y <= pairs_cap) {
        o.x = r.y - r.x;
        o.y = leaf_pairs[r.x].y;
        if (o.x > 1u) o.z = leaf_pairs[r.x + 1u].y;
        if (o.x > 2u) o2.x = leaf_pairs[r.x + 2u].y;
        if (o.x > 3u) o2.y = leaf_pairs[r.x + 3u].y;
        if (o.x > 4u) o2.z = leaf_pairs[r.x + 4u].y;
        if (o.x > 5u) o2.w = leaf_pairs[r.x + 5u].y;
    }
    rec[2u * c] = o;
    rec[2u * c + 1u] = o2;
}
void launch_tlg_leaf_records(hipStream_t s, const uint2* leaf_ranges, const uint2* leaf_pairs, uint32_t leaf_cells, uint32_t pairs_cap, uint4* rec)
{
    if (leaf_cells) hipLaunchKernelGGL(k_tlg_leaf_records, dim3((leaf_cells + TLG_BLOCK - 1) / TLG_BLOCK), dim3(TLG_BLOCK), 0, s, leaf_ranges, leaf_pairs, leaf_cells, pairs_cap, rec);
}
void launch_tlg_trace_records(hipStream_t s, const rt_grid_data& g, const rt_top_cell* top, uint32_t cells, TlgTopRec* rec, float4* axis_tab, uint32_t* too_fine)
{
    const uint32_t planes = g.resX + g.resY + g.resZ;
    hipLaunchKernelGGL(k_tlg_trace_records, dim3(((cells > planes ? cells : planes) + TLG_BLOCK - 1) / TLG_BLOCK), dim3(TLG_BLOCK), 0, s, g, top, cells, rec, axis_tab, too_fine);
}

// a load at a 32-bit byte offset from a buffer's base (the walk with its own records: every buffer it reads is below 4 GB, host check): the
// base stays in scalar registers and the address costs the one multiply / shift of the offset instead of 64-bit vector arithmetic per load.
// The load goes through the CONSTANT address space (nothing the walk reads is written during the launch): where the offset is the same in
// every lane -- made so with v_readfirstlane after a ballot, see the packet walk below -- the compiler fetches through the scalar cache into
// scalar registers (s_load), elsewhere it is the same global_load as before.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <typename T> struct LdOff {
    static __device__ __forceinline__ T at(const void* base, uint32_t byte_off)
    {
        typedef const __attribute__((address_space(4))) T cT;
        return *(cT*)((unsigned long long)base + byte_off);
    }
};
template <> struct LdOff<float4> {
    static __device__ __forceinline__ float4 at(const void* base, uint32_t byte_off)
    {
        const f4v v = LdOff<f4v>::at(base, byte_off);
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
template <> struct LdOff<uint4> {
    static __device__ __forceinline__ uint4 at(const void* base, uint32_t byte_off)
    {
        const u4v v = LdOff<u4v>::at(base, byte_off);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};
template <typename T> __device__ __forceinline__ T ld_off(const void* base, uint32_t byte_off) { return LdOff<T>::at(base, byte_off); }

// a * b + c on 24-bit a, b (cell indices and resolutions of the walk with its records: below 2^10 / 2^24).  Written out: the compiler, knowing
// nothing of the operands' ranges, turns one of each pair of __umul24 + add into v_mad_u64_u32.  _s: b in a scalar register.
__device__ __forceinline__ uint32_t mad24_s(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t mad24_v(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// x / y given r = RN(1 / y), for operands as described above
__device__ __forceinline__ float div_by(float x, float y, float r)
{
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-y, q0, x);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-y, q1, x);
    return __builtin_fmaf(e1, r, q1);
}

// the DDA set-up both levels share: TwoLevelGrid.h:436-500 (leaf level) and :575-640 (top level)
struct Dda {
    float nx, ny, nz;        // next[]
    float dtx, dty, dtz;     // dt[]
    int ix, iy, iz;          // idx[]
    float t0, tmaxmax;
};
// step[] and stop[] (TwoLevelGrid.h:470-490 / :610-630) are functions of the direction's sign and the resolution alone
// (d > 0: +1 / res, else -1 / -1) and the same at both levels: not kept in registers (18 VGPRs = one wave per SIMD), evaluated
// from these three bits when an index advances
struct DdaSigns { bool px, py, pz; };

__device__ __forceinline__ void dda_axis_setup(float o, float d, float lo, float hi, uint32_t res, float& tmin, float& tmax)
{
    const float a = 1.0f / d;            // the z axis divides the double literal 1.0 (:445,584): same value after rounding
    tmin = ((a >= 0.0f ? lo : hi) - o) * a;
    tmax = ((a >= 0.0f ? hi : lo) - o) * a;
}

__device__ __forceinline__ void dda_axis_finish(float p, float d, float lo, float hi, uint32_t res, float tmin, float tmax,
                                                float& next, float& dt, int& idx)
{
    float f = (p - lo) * (float)res / (hi - lo);
    const float top = (float)res - 1;
    f = RMAX(f, 0.0f); f = RMIN(f, top);              // clamp(v,l,h) = min(max(v,l),h), CLPortability.h:401
    idx = (int)f;
    dt = (tmax - tmin) / (float)res;
    if (d > 0.0f) {
        next = tmin + (float)(idx + 1) * dt;
    } else {
        next = d == 0.0f ? FLT_MAX : tmin + (float)(res - (uint32_t)idx) * dt;
    }
}

__device__ __forceinline__ void dda_setup(f3 o, f3 d, f3 lo, f3 hi, uint32_t rx, uint32_t ry, uint32_t rz, Dda& s)
{
    float txmin, txmax, tymin, tymax, tzmin, tzmax;
    dda_axis_setup(o.x, d.x, lo.x, hi.x, rx, txmin, txmax);
    dda_axis_setup(o.y, d.y, lo.y, hi.y, ry, tymin, tymax);
    dda_axis_setup(o.z, d.z, lo.z, hi.z, rz, tzmin, tzmax);
    s.t0 = RMIN(txmin, RMIN(tymin, tzmin));
    s.tmaxmax = RMAX(txmax, RMAX(tymax, tzmax));
    const bool inside = (lo.x <= o.x) && (o.x <= hi.x) && (lo.y <= o.y) && (o.y <= hi.y) && (lo.z <= o.z) && (o.z <= hi.z);
    f3 p = o;
    if (!inside) p = f3{ o.x + d.x * s.t0, o.y + d.y * s.t0, o.z + d.z * s.t0 };
    dda_axis_finish(p.x, d.x, lo.x, hi.x, rx, txmin, txmax, s.nx, s.dtx, s.ix);
    dda_axis_finish(p.y, d.y, lo.y, hi.y, ry, tymin, tymax, s.ny, s.dty, s.iy);
    dda_axis_finish(p.z, d.z, lo.z, hi.z, rz, tzmin, tzmax, s.nz, s.dtz, s.iz);
}

// `minimal = min(next[x], min(next[y], next[z])); while (minimal != next[axis]) axis++`; -1: no axis (all NaN), the
// ray ends with NO_CONTACT (FIXED reading; the reference runs off the array)
__device__ __forceinline__ int dda_axis(const Dda& s)
{
    const float minimal = RMIN(s.nx, RMIN(s.ny, s.nz));
    return minimal == s.nx ? 0 : (minimal == s.ny ? 1 : (minimal == s.nz ? 2 : -1));
}

// next[axis] += dt[axis]; idx[axis] += step[axis]; returns idx[axis] == stop[axis]
// (value selects on purpose: with branches the compiler turns next[] / dt[] / idx[] back into a scratch array indexed by axis)
__device__ __forceinline__ bool dda_advance(Dda& s, int axis, const DdaSigns& g, uint32_t rx, uint32_t ry, uint32_t rz)
{
    const bool a0 = axis == 0, a1 = axis == 1, a2 = !(a0 | a1);
    const float nx = s.nx + s.dtx, ny = s.ny + s.dty, nz = s.nz + s.dtz;
    s.nx = a0 ? nx : s.nx; s.ny = a1 ? ny : s.ny; s.nz = a2 ? nz : s.nz;
    s.ix += a0 ? (g.px ? 1 : -1) : 0; s.iy += a1 ? (g.py ? 1 : -1) : 0; s.iz += a2 ? (g.pz ? 1 : -1) : 0;
    const bool hx = s.ix == (g.px ? (int)rx : -1), hy = s.iy == (g.py ? (int)ry : -1), hz = s.iz == (g.pz ? (int)rz : -1);
    return (a0 & hx) | (a1 & hy) | (a2 & hz);
}

// -DRTMI_TLG_DIAG (experiment twin only, `make exp EXPSRC=tlg_kernels EXPFLAGS=-DRTMI_TLG_DIAG`): loop trip counts per WAVE (what the
// wave executes: the longest lane sets every loop) and per LANE (what the rays need), and the sizes of the leaf ranges met
#ifdef RTMI_TLG_DIAG
__device__ unsigned long long g_tlg_diag[96];
#define TLG_DIAG(W, L) do { const unsigned long long m__ = __ballot(1); if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m__) - 1)) { \
        atomicAdd(&g_tlg_diag[W], 1ull); atomicAdd(&g_tlg_diag[L], (unsigned long long)__popcll(m__)); } } while (0)
#define TLG_DIAG_HIST(n) atomicAdd(&g_tlg_diag[16u + ((n) < 63u ? (n) : 63u)], 1ull)
#else
#define TLG_DIAG(W, L)
#define TLG_DIAG_HIST(n)
#endif

typedef uint32_t mail_t;
#ifndef TLG_MAIL
#define TLG_MAIL 16u          // entries of a lane's mailbox (a power of two)
#endif
#ifndef TLG_AXIS_MAX
#define TLG_AXIS_MAX 512u
#endif
// |o| <= 2^30 and 2^-20 <= |d| <= 2^20 (false for NaN): with a grid that passed k_tlg_trace_records' check every operand of the short division
// sequences lies between 2^-90 and 2^90 or is a numerator so small that its quotient truncates to cell 0 either way
__device__ __forceinline__ bool tlg_moderate(float o, float d) { return fabsf(o) <= 0x1p30f && fabsf(d) >= 0x1p-20f && fabsf(d) <= 0x1p20f; }
__device__ __forceinline__ float rcp_near(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
// x / (float)res with r = RN(1 / (float)res): three operations give the IEEE quotient for every res <= 2048 and every |x| in [2^-90, 2^90]
// (enumerated: tools/micro/div_check.hip)
__device__ __forceinline__ float div_by_res(float x, float y, float r)
{
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-y, q0, x);
    return __builtin_fmaf(e0, r, q0);
}
// dda_setup for a top-level cell whose planes and resolution come from the walk's records.  The operations of dda_axis_setup / dda_axis_finish in
// their order; 1 / d per ray; the two divisions per axis by the short sequences; no d == 0 case (such rays do not come here).
__device__ __forceinline__ void dda_axis_fast(float o, float d, float a, float4 plane, float fres, float rres, float& tmin, float& tmax)
{
    const float lo = plane.x, hi = plane.y;
    tmin = ((a >= 0.0f ? lo : hi) - o) * a;
    tmax = ((a >= 0.0f ? hi : lo) - o) * a;
}
__device__ __forceinline__ void dda_finish_fast(float p, float d, float4 plane, int res, float fres, float rres, float tmin, float tmax, float& next, float& dt, int& idx)
{
    const float lo = plane.x;
    float f = div_by((p - lo) * fres, plane.w, plane.z);              // plane.w = hi - lo, the subtraction done once per plane
    const float top = fres - 1;
    f = __builtin_amdgcn_fmed3f(f, 0.0f, top);                       // min(max(f, 0), top) for a finite f and top >= 0 (a -0 gives cell 0 either way)
    idx = (int)f;
    dt = div_by_res(tmax - tmin, fres, rres);
    const int k = d > 0.0f ? idx + 1 : res - idx;                    // (float)(idx + 1) / (float)(res - (uint32_t)idx)
    next = tmin + (float)k * dt;
}
__device__ __forceinline__ void dda_setup_fast(f3 o, f3 d, f3 a, float4 px, float4 py, float4 pz, const TlgTopRec& r, Dda& s)
{
    float txmin, txmax, tymin, tymax, tzmin, tzmax;
    dda_axis_fast(o.x, d.x, a.x, px, r.fres[0], r.rcp[0], txmin, txmax);
    dda_axis_fast(o.y, d.y, a.y, py, r.fres[1], r.rcp[1], tymin, tymax);
    dda_axis_fast(o.z, d.z, a.z, pz, r.fres[2], r.rcp[2], tzmin, tzmax);
    // (one v_min3: the operands are finite, so it differs from the reference's two selects in the sign of a zero at most, which changes no cell --
    //  p = o + d * t0 enters only (p - lo) * res / (hi - lo), truncated and clamped to a cell index; tmaxmax is read at the top level only)
    s.t0 = __builtin_fminf(txmin, __builtin_fminf(tymin, tzmin));
    s.tmaxmax = 0.0f;
    const bool inside = (px.x <= o.x) && (o.x <= px.y) && (py.x <= o.y) && (o.y <= py.y) && (pz.x <= o.z) && (o.z <= pz.y);
    f3 p = o;
    if (!inside) p = f3{ o.x + d.x * s.t0, o.y + d.y * s.t0, o.z + d.z * s.t0 };
    dda_finish_fast(p.x, d.x, px, (int)(r.packed & 1023u), r.fres[0], r.rcp[0], txmin, txmax, s.nx, s.dtx, s.ix);
    dda_finish_fast(p.y, d.y, py, (int)((r.packed >> 10) & 1023u), r.fres[1], r.rcp[1], tymin, tymax, s.ny, s.dty, s.iy);
    dda_finish_fast(p.z, d.z, pz, (int)(r.packed >> 20), r.fres[2], r.rcp[2], tzmin, tzmax, s.nz, s.dtz, s.iz);
}

// One DDA step of the walk with its own records (TwoLevelGrid.h:542-545 / :660-676), every next[] finite.  The axis: the reference scans for
// the first axis whose next[] equals min(next[x], min(next[y], next[z])) -- x if next[x] <= both others, else y if next[y] <= next[z], else z:
// three compares instead of two selects and three equality tests.  next[axis] += dt[axis]; idx[axis] += step[axis] runs under the lanes of
// that axis (three masked pairs of additions instead of nine selects: the compiler would turn the masks into selects).  The ray leaves the
// grid when idx[axis] reaches res (step +1) or -1 (step -1): idx starts inside [0, res) and moves by one, so that is (unsigned)idx >= res.
__device__ __forceinline__ bool dda_step_fast(Dda& s, int sx, int sy, int sz, uint32_t rx, uint32_t ry, uint32_t rz)
{
    // (one block of assembly: the three compares write their lane masks where the scalar unit combines them -- m0 = x, m1 = y & ~x, m2 = the
    //  rest of exec -- and exec takes each in turn; handed three bools the compiler rebuilds every mask through a vector register)
    unsigned long long t0, t1, t2, saved;
    asm volatile("v_cmp_le_f32_e64 %[t0], %[nx], %[ny]\n\t"
                 "v_cmp_le_f32_e64 %[t1], %[nx], %[nz]\n\t"
                 "v_cmp_le_f32_e64 %[t2], %[ny], %[nz]\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 %[t0], %[t0], %[t1]\n\t"            // x
                 "s_andn2_b64 %[t1], %[t2], %[t0]\n\t"          // y and not x
                 "s_or_b64 %[t2], %[t0], %[t1]\n\t"
                 "s_andn2_b64 %[t2], %[sv], %[t2]\n\t"          // neither: z
                 "s_mov_b64 exec, %[t0]\n\t"
                 "v_add_f32 %[nx], %[nx], %[dx]\n\t"
                 "v_add_u32 %[ix], %[ix], %[sx]\n\t"
                 "s_mov_b64 exec, %[t1]\n\t"
                 "v_add_f32 %[ny], %[ny], %[dy]\n\t"
                 "v_add_u32 %[iy], %[iy], %[sy]\n\t"
                 "s_mov_b64 exec, %[t2]\n\t"
                 "v_add_f32 %[nz], %[nz], %[dz]\n\t"
                 "v_add_u32 %[iz], %[iz], %[sz]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [nx] "+v"(s.nx), [ny] "+v"(s.ny), [nz] "+v"(s.nz), [ix] "+v"(s.ix), [iy] "+v"(s.iy), [iz] "+v"(s.iz),
                   [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [sv] "=&s"(saved)
                 : [dx] "v"(s.dtx), [dy] "v"(s.dty), [dz] "v"(s.dtz), [sx] "v"(sx), [sy] "v"(sy), [sz] "v"(sz)
                 : "scc");
    return ((uint32_t)s.ix >= rx) | ((uint32_t)s.iy >= ry) | ((uint32_t)s.iz >= rz);
}

// ---- the packet walk (round 5, second part) -------------------------------------------------------------------------------------------
// The 64 rays of a wave are an 8x8 pixel block, and what they meet they mostly meet TOGETHER: on config 4 the lanes still walking stand in ONE
// top-level cell at 79 % of the wave's top-level steps, in one leaf cell at 83 % of its leaf steps, and 91 % of the leaf entries name the same
// triangle in every lane (counted by the diag twin, -DRTMI_TLG_DIAG).  Where that holds the cell's record, its planes, the leaf cell's record
// and the triangle come through the SCALAR cache into scalar registers -- no vector-memory instruction, no address arithmetic, no vector
// registers for them -- and the lanes compute exactly what they computed before, on their own rays, with scalar operands.  A wave takes this
// walk when all of its rays qualify for the short divisions (FAST) and have the same direction signs (then near / far plane, step[] and the
// count of cells to the far side are scalar as well); where the lanes disagree about the cell the step runs as before.
struct UniSigns { int mx, my, mz; };      // wave-uniform, in scalar registers: 0 where d > 0 on the axis, ~0 where d < 0

// dda_setup_fast with every property of the CELL in scalar registers: planes { lo, hi, RN(1 / (hi - lo)), hi - lo }, (float)res, RN(1 / res), res.
// Same operations on the same values, with three exceptions that give the same results: the near / far plane is chosen by the wave's sign
// (a >= 0 iff d > 0 for the rays that come here); t0 is one v_min3 (operands finite: it differs from the reference's two selects in the sign
// of a zero at most, which changes no cell: p = o + d * t0 enters only (p - lo) * res / (hi - lo), truncated and clamped to a cell index);
// the clamp to [0, res - 1] is applied to the converted integer (v_cvt_i32_f32 truncates towards zero and saturates: the same cell for every finite f).
__device__ __forceinline__ void dda_axis_uniform(float o, float d, float a, int neg, float lo, float hi, float& tmin, float& tmax)
{
    // near / far plane by the wave's sign: a constant of the walk's instance (one per direction octant)
    const float near = neg ? hi : lo, far = neg ? lo : hi;
    tmin = (near - o) * a;
    tmax = (far - o) * a;
}
__device__ __forceinline__ void dda_finish_uniform(float p, int neg, float lo, float rcp_w, float w, int res, float fres, float rres, float tmin, float tmax,
                                                   float& next, float& dt, int& idx)
{
    const float f = div_by((p - lo) * fres, w, rcp_w);
    // (the instructions themselves: v_cvt_i32_f32 saturates, where a C conversion out of range is undefined; res - 1 is scalar)
    asm("v_cvt_i32_f32 %0, %1\n\tv_med3_i32 %0, %0, 0, %2" : "=&v"(idx) : "v"(f), "s"(res - 1));
    dt = div_by_res(tmax - tmin, fres, rres);
    // d > 0: idx + 1, else res - idx  =  (idx ^ neg) + c with c = 1 / res + 1: scalar
    const int c = 1 + (res & neg);
    next = tmin + (float)((idx ^ neg) + c) * dt;
}
__device__ __forceinline__ void dda_setup_uniform(f3 o, f3 d, f3 a, UniSigns g, float4 px, float4 py, float4 pz, float fx, float fy, float fz,
                                                  float rx, float ry, float rz, int resx, int resy, int resz, Dda& s)
{
    float txmin, txmax, tymin, tymax, tzmin, tzmax;
    dda_axis_uniform(o.x, d.x, a.x, g.mx, px.x, px.y, txmin, txmax);
    dda_axis_uniform(o.y, d.y, a.y, g.my, py.x, py.y, tymin, tymax);
    dda_axis_uniform(o.z, d.z, a.z, g.mz, pz.x, pz.y, tzmin, tzmax);
    s.t0 = __builtin_fminf(txmin, __builtin_fminf(tymin, tzmin));
    s.tmaxmax = 0.0f;                                                // (read at the top level only)
    const bool inside = (px.x <= o.x) && (o.x <= px.y) && (py.x <= o.y) && (o.y <= py.y) && (pz.x <= o.z) && (o.z <= pz.y);
    f3 p = o;
    if (!inside) p = f3{ o.x + d.x * s.t0, o.y + d.y * s.t0, o.z + d.z * s.t0 };
    dda_finish_uniform(p.x, g.mx, px.x, px.z, px.w, resx, fx, rx, txmin, txmax, s.nx, s.dtx, s.ix);
    dda_finish_uniform(p.y, g.my, py.x, py.z, py.w, resy, fy, ry, tymin, tymax, s.ny, s.dty, s.iy);
    dda_finish_uniform(p.z, g.mz, pz.x, pz.z, pz.w, resz, fz, rz, tzmin, tzmax, s.nz, s.dtz, s.iz);
}
// ... and for primary rays with the planes relative to the camera (TlgCamPlane): (near - o) and (far - o) come from the table -- the same
// subtractions, done once per camera position -- and `inside` is the AND of three flags: scalar, so p = o + d * t0 is computed where it is used
// what the set-up reads of a camera plane: { lo, rcp_w, w, inside } and (near - o, far - o) -- the two words at 16 + 4 * (d < 0)
struct CamPlaneS { float lo, rcp_w, w; uint32_t inside; float near_o, far_o; };
__device__ __forceinline__ CamPlaneS ld_cam_plane(const void* base, uint32_t byte_off, int neg)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f4v a = LdOff<f4v>::at(base, byte_off);
    const f2v b = LdOff<f2v>::at(base, byte_off + 16u + ((uint32_t)neg & 4u));
    return CamPlaneS{ a.x, a.y, a.z, __float_as_uint(a.w), b.x, b.y };
}
__device__ __forceinline__ void dda_setup_uniform_cam(f3 o, f3 d, f3 a, UniSigns g, const CamPlaneS& cx, const CamPlaneS& cy, const CamPlaneS& cz,
                                                      float fx, float fy, float fz, float rx, float ry, float rz, int resx, int resy, int resz, Dda& s)
{
    const float txmin = cx.near_o * a.x, txmax = cx.far_o * a.x, tymin = cy.near_o * a.y, tymax = cy.far_o * a.y, tzmin = cz.near_o * a.z, tzmax = cz.far_o * a.z;
    s.t0 = __builtin_fminf(txmin, __builtin_fminf(tymin, tzmin));
    s.tmaxmax = 0.0f;
    f3 p = f3{ o.x + d.x * s.t0, o.y + d.y * s.t0, o.z + d.z * s.t0 };
    // (a scalar condition, true in one cell per camera: a branch -- the empty statement keeps the compiler from turning it into three selects)
    if ((cx.inside & cy.inside & cz.inside) != 0u) { asm volatile("; origin inside this cell"); p = o; }
    dda_finish_uniform(p.x, g.mx, cx.lo, cx.rcp_w, cx.w, resx, fx, rx, txmin, txmax, s.nx, s.dtx, s.ix);
    dda_finish_uniform(p.y, g.my, cy.lo, cy.rcp_w, cy.w, resy, fy, ry, tymin, tymax, s.ny, s.dty, s.iy);
    dda_finish_uniform(p.z, g.mz, cz.lo, cz.rcp_w, cz.w, resz, fz, rz, tzmin, tzmax, s.nz, s.dtz, s.iz);
}
// dda_step_fast with step[] as constants of the instruction (the packet walk is compiled once per direction octant: +1 / -1 are inline operands)
template <int SX, int SY, int SZ>
__device__ __forceinline__ bool dda_step_uniform(Dda& s, uint32_t rx, uint32_t ry, uint32_t rz)
{
    unsigned long long t0, t1, t2, saved;
    asm volatile("v_cmp_le_f32_e64 %[t0], %[nx], %[ny]\n\t"
                 "v_cmp_le_f32_e64 %[t1], %[nx], %[nz]\n\t"
                 "v_cmp_le_f32_e64 %[t2], %[ny], %[nz]\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 %[t0], %[t0], %[t1]\n\t"
                 "s_andn2_b64 %[t1], %[t2], %[t0]\n\t"
                 "s_or_b64 %[t2], %[t0], %[t1]\n\t"
                 "s_andn2_b64 %[t2], %[sv], %[t2]\n\t"
                 "s_mov_b64 exec, %[t0]\n\t"
                 "v_add_f32 %[nx], %[nx], %[dx]\n\t"
                 "v_add_u32 %[ix], %[sx], %[ix]\n\t"
                 "s_mov_b64 exec, %[t1]\n\t"
                 "v_add_f32 %[ny], %[ny], %[dy]\n\t"
                 "v_add_u32 %[iy], %[sy], %[iy]\n\t"
                 "s_mov_b64 exec, %[t2]\n\t"
                 "v_add_f32 %[nz], %[nz], %[dz]\n\t"
                 "v_add_u32 %[iz], %[sz], %[iz]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [nx] "+v"(s.nx), [ny] "+v"(s.ny), [nz] "+v"(s.nz), [ix] "+v"(s.ix), [iy] "+v"(s.iy), [iz] "+v"(s.iz),
                   [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [sv] "=&s"(saved)
                 : [dx] "v"(s.dtx), [dy] "v"(s.dty), [dz] "v"(s.dtz), [sx] "n"(SX), [sy] "n"(SY), [sz] "n"(SZ)
                 : "scc");
    return ((uint32_t)s.ix >= rx) | ((uint32_t)s.iy >= ry) | ((uint32_t)s.iz >= rz);
}

// the kernel's (only) argument read from the kernel-argument segment at the point of use (an opaque pointer: the loads can neither be merged
// with the ones at the kernel's start nor be moved up)
__device__ __forceinline__ const TlgTraceArgs* tlg_args_again()
{
    const TlgTraceArgs* p = (const TlgTraceArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; kernel arguments again" : "+s"(p));
    return p;
}
// The end of triangleIntersect + :528 -- `flag` and the update of the cell's closest hit -- as a chain of v_cmpx: every compare narrows exec to the
// lanes that passed so far, the update is two moves under what is left.  The same seven ordered compares as the C expression
//   |det| >= FLT_EPSILON & u < 1 & u >= 0 & u + v <= 1 & v >= 0 & t > 0 & t < cell_t
// without the six scalar ANDs of their lane masks and the selects (this kernel's time follows the number of instructions it issues, scalar ones
// included).  _s: the triangle id in a scalar register (the packet walk's leaf cells).
// (second form: v = (d . qvec) / det, t and the compares that need them stand BEHIND a branch of the wave -- a triangle most of whose tests fail fails
//  them at |det| or u for every lane that tested it, and the wave then skips thirteen instructions.  The arithmetic inside is f3_dot's, operation for
//  operation: x * x' + y * y' first, then + z * z', every operation rounded on its own.)
#define TLG_ACCEPT_ASM(Q_CONSTRAINT)                                                                                                            \
    unsigned long long saved;                                                                                                                   \
    float t1, t2;                                                                                                                               \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                                                                     \
                 "v_cmpx_ge_f32_e64 vcc, |%[det]|, %[eps]\n\t"                                                                                   \
                 "v_cmpx_gt_f32_e32 vcc, 1.0, %[u]\n\t"                                                                                          \
                 "v_cmpx_le_f32_e32 vcc, 0, %[u]\n\t"                                                                                            \
                 "s_cbranch_execz 1f\n\t"                                                                                                        \
                 "v_mul_f32 %[t1], %[qx], %[dx]\n\t"                                                                                             \
                 "v_mul_f32 %[t2], %[qy], %[dy]\n\t"                                                                                             \
                 "v_add_f32 %[t1], %[t1], %[t2]\n\t"                                                                                             \
                 "v_mul_f32 %[t2], %[qz], %[dz]\n\t"                                                                                             \
                 "v_add_f32 %[t1], %[t1], %[t2]\n\t"                                                                                             \
                 "v_mul_f32 %[t1], %[t1], %[inv]\n\t"                                                                                            \
                 "v_add_f32 %[t2], %[u], %[t1]\n\t"                                                                                              \
                 "v_cmpx_ge_f32_e32 vcc, 1.0, %[t2]\n\t"                                                                                         \
                 "v_cmpx_le_f32_e32 vcc, 0, %[t1]\n\t"                                                                                           \
                 "v_mul_f32 %[t1], %[e2q], %[inv]\n\t"                                                                                           \
                 "v_cmpx_lt_f32_e32 vcc, 0, %[t1]\n\t"                                                                                           \
                 "v_cmpx_gt_f32_e32 vcc, %[ct], %[t1]\n\t"                                                                                       \
                 "v_mov_b32 %[ct], %[t1]\n\t"                                                                                                    \
                 "v_mov_b32 %[ctri], %[tri]\n"                                                                                                   \
                 "1:\n\t"                                                                                                                        \
                 "s_mov_b64 exec, %[sv]"                                                                                                          \
                 : [ct] "+v"(cell_t), [ctri] "+v"(cell_tri), [sv] "=&s"(saved), [t1] "=&v"(t1), [t2] "=&v"(t2)                                    \
                 : [det] "v"(det), [eps] "s"(eps), [u] "v"(u), [inv] "v"(inv_det), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z),                   \
                   [qx] Q_CONSTRAINT(qvec.x), [qy] Q_CONSTRAINT(qvec.y), [qz] Q_CONSTRAINT(qvec.z), [e2q] Q_CONSTRAINT(e2q), [tri] Q_CONSTRAINT(tri) \
                 : "vcc", "scc")
__device__ __forceinline__ void tlg_accept_s(float det, float inv_det, float u, f3 d, f3 qvec, float e2q, uint32_t tri, float& cell_t, uint32_t& cell_tri)
{
    const float eps = FLT_EPSILON;
    TLG_ACCEPT_ASM("s");
}
__device__ __forceinline__ void tlg_accept_v(float det, float inv_det, float u, f3 d, f3 qvec, float e2q, uint32_t tri, float& cell_t, uint32_t& cell_tri)
{
    const float eps = FLT_EPSILON;
    TLG_ACCEPT_ASM("v");
}

// The walk of one ray (tlg_generate_contact, TwoLevelGrid.h:548-678).  FAST: every ray of this wave and the grid qualify for the walk's own
// records (see k_tlg_trace_records): the leaf-level set-up takes its cell from a 32-byte record and its planes from a table in LDS, divides by
// the sequences above, and the triangle test takes 1 / det from v_rcp_f32 + one Newton step (equal to the IEEE quotient for every operand
// between 2^-90 and 2^90, tools/micro/div_check.hip; |det| >= FLT_EPSILON is what the test asks for anyway).  Same values, fewer instructions.
template <int PRIMARY, int FAST, int OCT = 0>         // FAST = 2: the packet walk (FAST = 1 and the wave's rays have the same direction signs: bit k of OCT = d < 0 on axis k)
__device__ __forceinline__ void tlg_walk(const TlgTraceArgs& a, const f3 o, const f3 d, mail_t (*s_mailbox)[TLG_BLOCK], const float4* s_axis,
                                         float& best_t, uint32_t& best_tri)
{
    const rt_grid_data& g = a.grid;
    const f3 glo = f3{ g.box_lo[0], g.box_lo[1], g.box_lo[2] }, ghi = f3{ g.box_hi[0], g.box_hi[1], g.box_hi[2] };

    const DdaSigns sg = DdaSigns{ d.x > 0.0f, d.y > 0.0f, d.z > 0.0f };
    const int step_x = sg.px ? 1 : -1, step_y = sg.py ? 1 : -1, step_z = sg.pz ? 1 : -1;         // step[] of :470-490 / :610-630
    // FAST = 2: the signs are the instance's (the kernel picks it by the wave's ballots): 0 where d > 0 on the axis, ~0 where d < 0 -- constants, so
    // near / far plane, step[], the count of cells to the far side and the word offsets of the camera's plane table cost neither registers nor instructions
    const UniSigns us = UniSigns{ (OCT & 1) ? -1 : 0, (OCT & 2) ? -1 : 0, (OCT & 4) ? -1 : 0 };
    constexpr int ustep_x = (OCT & 1) ? -1 : 1, ustep_y = (OCT & 2) ? -1 : 1, ustep_z = (OCT & 4) ? -1 : 1;          // step[]
    const f3 inv_d = f3{ 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };          // (what dda_axis_setup divides per cell; the compiler hoists it there as well)
    // where the camera's planes of the three axes start in toprec's buffer (bytes)
    const uint32_t cam_x = a.cam_off, cam_y = a.cam_off + g.resX * 32u, cam_z = a.cam_off + (g.resX + g.resY) * 32u;
#ifdef RTMI_TLG_DIAG
    if (FAST) {       // (86) waves all of whose rays have the same direction signs; (88) all waves taking this walk
        const unsigned long long ax = __ballot(sg.px), ay = __ballot(sg.py), az = __ballot(sg.pz), all = __ballot(1);
        if ((ax == 0 || ax == all) && (ay == 0 || ay == all) && (az == 0 || az == all)) TLG_DIAG(86, 87);
        TLG_DIAG(88, 89);
    }
#endif
    bool done;
    Dda T;
    dda_setup(o, d, glo, ghi, g.resX, g.resY, g.resZ, T);
    done = T.t0 > T.tmaxmax;
    while (!done) {
        TLG_DIAG(0, 1);
        const int tix = T.ix, tiy = T.iy, tiz = T.iz;
        // (at most 512 planes with the walk's records: every top-level resolution is below 512)
        const uint32_t cell = FAST ? mad24_s(mad24_s((uint32_t)tiz, g.resY, (uint32_t)tiy), g.resX, (uint32_t)tix)
                                   : (uint32_t)tiz * g.resX * g.resY + (uint32_t)tiy * g.resX + (uint32_t)tix;
#ifdef RTMI_TLG_DIAG
        if (FAST) {       // (80) top-level steps whose active lanes all stand in ONE cell; (82) ... with all 64 lanes still walking
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell);
            if (__builtin_amdgcn_ballot_w64(cell != c0) == 0ull) { TLG_DIAG(80, 81); if (__popcll(__ballot(1)) == 64) TLG_DIAG(82, 83); }
        }
#endif
        float cell_t = FLT_MAX;
        uint32_t cell_tri = 0xFFFFFFFFu;
        // one entry of a leaf cell: mailbox, then the triangle test.  (`tri` the same in every lane -- the packet walk's leaf cells -- : the
        // triangle comes in scalar registers; the mailbox stays the lane's own, for what a lane tested before differs from lane to lane)
        auto test_entry = [&](const uint32_t tri, auto tri_is_scalar) __attribute__((always_inline)) {
#ifdef RTMI_TLG_DIAG
            if (FAST) { const uint32_t t0_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)tri); if (__builtin_amdgcn_ballot_w64(tri != t0_) == 0ull) TLG_DIAG(90, 91); TLG_DIAG(92, 93); }
#endif
            mail_t* slot = &s_mailbox[tri & (TLG_MAIL - 1u)][threadIdx.x];
            const mail_t tag = (mail_t)tri;                   // (the whole id: no shift; an empty slot holds ~0, no triangle)
            if (*slot == tag) return;                  // tested in an earlier cell: a miss, again
            *slot = tag;
            TLG_DIAG(8, 9);
            // triangleIntersect, Triangle.h:66-107 (edges precomputed with the same subtractions); for primary rays also
            // what depends on the origin and the triangle alone -- tvec, qvec, edge2 . qvec -- once per camera position
            f3 edge1, edge2, tvec, qvec;
            float e2q;
            if (PRIMARY && FAST && (FAST == 2 || a.tricam)) {          // (the packet walk of primary rays is launched with the records only)
                const uint32_t at = tri * 64u;
                const float4 w0 = ld_off<float4>(a.tricam, at), w1 = ld_off<float4>(a.tricam, at + 16u), w2 = ld_off<float4>(a.tricam, at + 32u);
                edge1 = f3{ w0.x, w0.y, w0.z }; edge2 = f3{ w0.w, w1.x, w1.y }; tvec = f3{ w1.z, w1.w, w2.x }; qvec = f3{ w2.y, w2.z, w2.w };
                e2q = ld_off<float>(a.tricam, at + 48u);
            } else if (PRIMARY && a.tricam) {
                const float4* p = reinterpret_cast<const float4*>(a.tricam + tri);
                const float4 w0 = p[0], w1 = p[1], w2 = p[2];
                edge1 = f3{ w0.x, w0.y, w0.z }; edge2 = f3{ w0.w, w1.x, w1.y }; tvec = f3{ w1.z, w1.w, w2.x }; qvec = f3{ w2.y, w2.z, w2.w };
                e2q = p[3].x;
            } else {
                const f3 vert0 = f3_from4(FAST ? ld_off<float4>(a.v0, tri * 16u) : a.v0[tri]);
                edge1 = f3_from4(FAST ? ld_off<float4>(a.e1, tri * 16u) : a.e1[tri]); edge2 = f3_from4(FAST ? ld_off<float4>(a.e2, tri * 16u) : a.e2[tri]);
                tvec = f3_sub(o, vert0);
                qvec = f3_cross(tvec, edge1);
                e2q = f3_dot(edge2, qvec);
            }
            const f3 pvec = f3_cross(d, edge2);
            const float det = f3_dot(edge1, pvec);
            const float inv_det = FAST ? rcp_near(det) : 1.0f / det;
            const float u = f3_dot(tvec, pvec) * inv_det;
            if (FAST) {
                // v = (d . qvec) * inv_det; t = e2q * inv_det ("* flag" folded into the test);
                // flag = |det| >= FLT_EPSILON & u < 1 & u >= 0 & u + v <= 1 & v >= 0; if (flag & t > 0 & t < cell_t) { cell_t = t; cell_tri = tri; }   (:528)
                // (for a triangle in scalar registers of a PRIMARY wave qvec and e2q are scalar; a ray buffer's are the lane's own)
                if (decltype(tri_is_scalar)::value && PRIMARY) tlg_accept_s(det, inv_det, u, d, qvec, e2q, tri, cell_t, cell_tri);
                else tlg_accept_v(det, inv_det, u, d, qvec, e2q, tri, cell_t, cell_tri);
            } else {
                const float v = f3_dot(d, qvec) * inv_det;
                const float t = e2q * inv_det;                         // "* flag" folded into the test
                bool flag = fabsf(det) >= FLT_EPSILON;
                flag = flag & (u < 1.0f) & (u >= 0.0f);
                flag = flag & ((u + v) <= 1.0f) & (v >= 0.0f);
                if (flag & (t > 0.0f) & (t < cell_t)) { cell_t = t; cell_tri = tri; }   // :528
            }
        };
        // the entries of one leaf cell, from the walk's leaf-cell record
        auto walk_leaf = [&](const uint32_t leaf) __attribute__((always_inline)) {
            const uint4 lr = ld_off<uint4>(a.leafrec, leaf * 32u);
            TLG_DIAG_HIST(lr.x);
            if (FAST == 2) {
                // (the packet walk's lane-by-lane leaf cells: the first six entries from the record as well -- no dependent fetch of a pair per entry, and
                //  the pair array's pointer is read again from the kernel arguments for the rare seventh entry, not held in scalar registers through the walk)
                const uint4 lq = ld_off<uint4>(a.leafrec, leaf * 32u + 16u);
                if (lr.x > 0u) { test_entry(lr.y, std::false_type());
                if (lr.x > 1u) { test_entry(lr.z, std::false_type());
                if (lr.x > 2u) { test_entry(lq.x, std::false_type());
                if (lr.x > 3u) { test_entry(lq.y, std::false_type());
                if (lr.x > 4u) { test_entry(lq.z, std::false_type());
                if (lr.x > 5u) { test_entry(lq.w, std::false_type());
                    const uint2* pairs = tlg_args_again()->leaf_pairs;
                    for (uint32_t i = 6u; i < lr.x; i++) test_entry(ld_off<uint32_t>(pairs, (lr.w + i) * 8u + 4u), std::false_type());
                } } } } } }
                return;
            }
            uint32_t tri_next = lr.y;
            for (uint32_t i = 0; i < lr.x; i++) {
                TLG_DIAG(6, 7);
                const uint32_t tri = tri_next;
                // (entry 1 comes with the record; later ones are fetched one entry ahead, while this triangle is tested)
                if (i + 1u < lr.x) tri_next = i == 0u ? lr.z : ld_off<uint32_t>(a.leaf_pairs, (lr.w + i) * 8u + 12u);
                test_entry(tri, std::false_type());
            }
        };
        const uint32_t c0 = FAST == 2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cell) : 0u;
        if (FAST == 2 && __builtin_amdgcn_ballot_w64(cell != c0) == 0ull) {
            {
                // ---- every lane still walking stands in cell c0: its record and planes in scalar registers ----
                // (record and planes requested together, before the cell is known to be non-empty and to need a set-up -- nearly every cell a ray meets
                //  does: otherwise three trips through the scalar cache in a row per cell)
                const uint32_t ux = (uint32_t)__builtin_amdgcn_readfirstlane(tix), uy = (uint32_t)__builtin_amdgcn_readfirstlane(tiy),
                               uz = (uint32_t)__builtin_amdgcn_readfirstlane(tiz);
                const float4 r0 = ld_off<float4>(a.toprec, c0 * 32u), r1 = ld_off<float4>(a.toprec, c0 * 32u + 16u);
                float4 px, py, pz;
                CamPlaneS cx, cy, cz;
                if (PRIMARY) {
                    cx = ld_cam_plane(a.toprec, cam_x + ux * 32u, us.mx); cy = ld_cam_plane(a.toprec, cam_y + uy * 32u, us.my); cz = ld_cam_plane(a.toprec, cam_z + uz * 32u, us.mz);
                    // (read-only operands: the values have to be there, no copies made)
                    asm volatile("" :: "s"(r0.x), "s"(r0.w), "s"(r1.x), "s"(r1.w), "s"(cx.lo), "s"(cx.far_o), "s"(cy.lo), "s"(cy.far_o), "s"(cz.lo), "s"(cz.far_o));
                } else {
                    px = ld_off<float4>(a.axis_tab, ux * 16u); py = ld_off<float4>(a.axis_tab, (g.resX + uy) * 16u); pz = ld_off<float4>(a.axis_tab, (g.resX + g.resY + uz) * 16u);
                    asm volatile("" :: "s"(r0.x), "s"(r0.w), "s"(r1.x), "s"(r1.w), "s"(px.x), "s"(px.w), "s"(py.x), "s"(py.w), "s"(pz.x), "s"(pz.w));
                }
                const uint32_t packed = __float_as_uint(r1.w), first_leaf = __float_as_uint(r0.w);
                // one leaf cell of this top-level cell: its entries from scalar registers where the lanes stand in the same one, else lane by lane
                auto leaf_cell = [&](const uint32_t leaf) __attribute__((always_inline)) {
                    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf);
                    if (__builtin_amdgcn_ballot_w64(leaf != l0) == 0ull) {
                        // ---- ... and in ONE leaf cell: its record and, entry by entry, its triangles in scalar registers ----
#ifdef RTMI_TLG_DIAG
                        TLG_DIAG(84, 85);
#endif
                        // (up to six entries straight from the record, no loop: an entry costs its mailbox look-up and one scalar compare + branch)
                        const uint4 lr = ld_off<uint4>(a.leafrec, l0 * 32u), lq = ld_off<uint4>(a.leafrec, l0 * 32u + 16u);
                        if (lr.x > 0u) { test_entry(lr.y, std::true_type());
                        if (lr.x > 1u) { test_entry(lr.z, std::true_type());
                        if (lr.x > 2u) { test_entry(lq.x, std::true_type());
                        if (lr.x > 3u) { test_entry(lq.y, std::true_type());
                        if (lr.x > 4u) { test_entry(lq.z, std::true_type());
                        if (lr.x > 5u) { test_entry(lq.w, std::true_type());
                            const uint2* pairs = tlg_args_again()->leaf_pairs;       // (rare: the pointer is not held in scalar registers through the walk)
                            for (uint32_t i = 6u; i < lr.x; i++) test_entry(ld_off<uint32_t>(pairs, (lr.w + i) * 8u + 4u), std::true_type());
                        } } } } } }
                    } else {
                        walk_leaf(leaf);
                    }
                };
                if (packed == (1u | (1u << 10) | (1u << 20))) {
                    // a leaf grid of ONE cell needs no DDA: its cell is processed, and whichever axis the step would take leaves the grid (:542-545)
                    TLG_DIAG(2, 3);
                    TLG_DIAG(4, 5);
                    leaf_cell(first_leaf);
                } else if (packed != 0u) {
                    TLG_DIAG(2, 3);
                    const uint32_t rx = packed & 1023u, ry = (packed >> 10) & 1023u, rz = packed >> 20;
                    Dda Lf;
                    if (PRIMARY) dda_setup_uniform_cam(o, d, inv_d, us, cx, cy, cz, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, (int)rx, (int)ry, (int)rz, Lf);
                    else dda_setup_uniform(o, d, inv_d, us, px, py, pz, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, (int)rx, (int)ry, (int)rz, Lf);
                    for (;;) {                   // processTopLevelCell, TwoLevelGrid.h:419-545
                        TLG_DIAG(4, 5);
                        leaf_cell(mad24_s(mad24_s((uint32_t)Lf.iz, ry, (uint32_t)Lf.iy), rx, (uint32_t)Lf.ix) + first_leaf);
                        if (cell_tri != 0xFFFFFFFFu) break;                       // :539-540
                        if (dda_step_uniform<ustep_x, ustep_y, ustep_z>(Lf, rx, ry, rz)) break;   // :542-545
                    }
                }
                if (cell_tri != 0xFFFFFFFFu) { best_t = cell_t; best_tri = cell_tri; break; }   // :668-669
            }
        }
        else {
        rt_top_cell tc;
        TlgTopRec rec;
        if (FAST) {
            const float4 r0 = ld_off<float4>(a.toprec, cell * 32u), r1 = ld_off<float4>(a.toprec, cell * 32u + 16u);
            // (both halves and the planes are fetched BEFORE the cell is known to be non-empty -- nearly every cell a ray meets is: left to itself the
            //  compiler fetches the last word, waits, branches and only then fetches the rest: two trips to the L2 in a row per cell)
            asm volatile("" :: "v"(r0.x), "v"(r0.w), "v"(r1.x), "v"(r1.w));          // (read-only operands: the values have to be there, no copies made)
            rec.fres[0] = r0.x; rec.fres[1] = r0.y; rec.fres[2] = r0.z; rec.first_leaf = __float_as_uint(r0.w);
            rec.rcp[0] = r1.x; rec.rcp[1] = r1.y; rec.rcp[2] = r1.z; rec.packed = __float_as_uint(r1.w);
            tc.resX = rec.packed & 1023u; tc.resY = (rec.packed >> 10) & 1023u; tc.resZ = rec.packed >> 20; tc.firstLeafIdx = rec.first_leaf;
        } else {
            tc = a.top[cell];
        }
        if (FAST ? rec.packed != 0u : !(tc.resX == 0 || tc.resY == 0 || tc.resZ == 0)) {
            TLG_DIAG(2, 3);
            // A leaf grid of one cell needs no DDA: its cell is processed, and whichever axis the step would take leaves the grid (:542-545).  Taken
            // when every lane of the wave stands in such a top-level cell (11 % of the wave's top-level steps on config 4); a lane among others runs
            // through the general code below, to the same effect.
            const bool one_cell = FAST == 1 && __builtin_amdgcn_ballot_w64(rec.packed != (1u | (1u << 10) | (1u << 20))) == 0ull;
            if (one_cell) {
                walk_leaf(rec.first_leaf);
            } else {
            // processTopLevelCell, TwoLevelGrid.h:419-545
            Dda Lf;
            if (FAST) {
                const float4 px = s_axis[tix], py = s_axis[g.resX + (uint32_t)tiy], pz = s_axis[g.resX + g.resY + (uint32_t)tiz];     // { lo, hi, 1 / (hi - lo), hi - lo }
                dda_setup_fast(o, d, inv_d, px, py, pz, rec, Lf);
            } else {
                f3 clo, chi;
                clo.x = glo.x + (float)tix * g.stepX; clo.y = glo.y + (float)tiy * g.stepY; clo.z = glo.z + (float)tiz * g.stepZ;
                chi.x = clo.x + g.stepX; chi.y = clo.y + g.stepY; chi.z = clo.z + g.stepZ;
                dda_setup(o, d, clo, chi, tc.resX, tc.resY, tc.resZ, Lf);
            }
            for (;;) {
                TLG_DIAG(4, 5);
                const int laxis = FAST ? 0 : dda_axis(Lf);
                // (with the walk's records every leaf resolution is below 1024 and every index below it: the same sum by 24-bit multiply-adds)
                const uint32_t leaf = FAST ? mad24_v(mad24_v((uint32_t)Lf.iz, tc.resY, (uint32_t)Lf.iy), tc.resX, (uint32_t)Lf.ix) + tc.firstLeafIdx
                                           : (uint32_t)Lf.iz * tc.resX * tc.resY + (uint32_t)Lf.iy * tc.resX + (uint32_t)Lf.ix + tc.firstLeafIdx;
                if (FAST) {
#ifdef RTMI_TLG_DIAG
                    {   // (84) leaf-level steps whose active lanes all stand in ONE leaf cell
                        const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf);
                        if (__builtin_amdgcn_ballot_w64(leaf != l0) == 0ull) TLG_DIAG(84, 85);
                    }
#endif
                    walk_leaf(leaf);
                } else {
                    const uint2 range = a.leaf_ranges[leaf];
                    TLG_DIAG_HIST(range.y - range.x);
                    // (the next pair id is fetched while this triangle is tested: one dependent load less per entry)
                    uint32_t tri_next = range.x < range.y ? a.leaf_pairs[range.x].y : 0u;
                    for (uint32_t i = range.x; i < range.y; i++) {
                        TLG_DIAG(6, 7);
                        const uint32_t tri = tri_next;
                        if (i + 1u < range.y) tri_next = a.leaf_pairs[i + 1u].y;
                        test_entry(tri, std::false_type());
                    }
                }
                if (cell_tri != 0xFFFFFFFFu) break;                       // :539-540
                if (FAST == 2) { if (dda_step_uniform<ustep_x, ustep_y, ustep_z>(Lf, tc.resX, tc.resY, tc.resZ)) break; }
                else if (FAST) { if (dda_step_fast(Lf, step_x, step_y, step_z, tc.resX, tc.resY, tc.resZ)) break; }
                else {
                    if (laxis < 0) break;
                    if (dda_advance(Lf, laxis, sg, tc.resX, tc.resY, tc.resZ)) break;   // :542-545
                }
            }
            }
            if (cell_tri != 0xFFFFFFFFu) { best_t = cell_t; best_tri = cell_tri; break; }   // :668-669
        }
        }
        // :660-676 -- the axis of the smallest next[] (chosen before the cell is processed in the reference; nothing in between changes
        // next[]), `next[axis] += dt[axis]; idx[axis] += step[axis]; if (idx[axis] == stop[axis]) break`
        if (FAST == 2) { if (dda_step_uniform<ustep_x, ustep_y, ustep_z>(T, g.resX, g.resY, g.resZ)) break; }
        else if (FAST) { if (dda_step_fast(T, step_x, step_y, step_z, g.resX, g.resY, g.resZ)) break; }
        else {
            const int taxis = dda_axis(T);
            if (taxis < 0) break;
            if (dda_advance(T, taxis, sg, g.resX, g.resY, g.resZ)) break;
        }
    }

}

// Registers set this kernel's occupancy (round 3's counters: a wave issues for 35 % of its life and is parked on loads for 43 %; five
// waves per SIMD kept the vector unit 50 % busy).  Six waves per SIMD = 80 VGPRs (86 before), reached by (a) step[] / stop[] as
// functions of three sign bits (DdaSigns) and (b) the pixel located again for the store instead of kept: config 4's trace 3.57 ->
// 3.36-3.41 ms.  Measured beyond that and not kept: seven waves (72 VGPRs: the top level's next[] / dt[] / idx[] in LDS, [k][lane]
// with the axis as the index, 16-bit mailbox tags so that nine blocks fit a CU's LDS) 3.35-3.47; eight (spills) 3.77; fetching the next
// top-level cell / the next leaf range before the current one is worked on 3.62 / 3.63 (the copies of the DDA state cost what the
// overlap gains) -- only the next PAIR id is fetched early (3.56 -> 3.48).
// Round 5: with the walk's own records the kernel needs 72 VGPRs and no spill in its loops: SEVEN waves per SIMD (the plane table is dynamic LDS
// sized by the grid, so 16 KB of mailbox + ~4 KB of planes leave room for seven blocks per CU): 2.71 -> 2.62 ms then, 2.29 at the end of the
// round; eight waves (64 VGPRs, scalar registers spilled into lanes) 2.43, six 2.50; 16-bit mailbox tags 2.26-2.28 (noise, and a limit on the
// triangle count); the top-level step taken early with a warming fetch of the next cell's record 2.34.  DESIGN.md section 4.6.
// Round 5, second part (the packet walk): 2.29 -> 1.86 ms at the same seven waves (72 VGPRs; the lane-per-cell half of a step sets them).  The launch
// time followed the TOTAL number of instructions issued, scalar ones included, at 0.95-0.97 ns per million through the steps
// (profiles/r05_tlg_instruction_counts.json): 2 392 M -> 1 817 M.  Measured on the way and not kept: eight waves 2.06-2.21 (spills), six 2.27; other
// scheduler strategies (max-ilp 2.06, the default 2.00 = max-memory-clause); workgroups of 64 / 128 threads with the planes from the global table instead of
// LDS 2.15-2.17 against 2.15 then; the walk as a loop of the wave with the cell's indices in scalar registers (fewer vector, more scalar
// instructions: 2.27 against 2.15); a 64-entry mailbox would skip 5 % more tests (counted by the diag twin) for four times the LDS; the mailbox slots
// of a leaf cell's entries read ahead (a second ladder of compares: 2.07 against 2.04).
#ifndef TLG_WAVES
#define TLG_WAVES 7
#endif
template <int PRIMARY>
__global__ __launch_bounds__(TLG_BLOCK) __attribute__((amdgpu_waves_per_eu(TLG_WAVES, TLG_WAVES))) void k_tlg_trace(TlgTraceArgs a)
{
    // Mailbox: a triangle lies in every leaf cell it overlaps, so a ray marching through the grid meets it again and again
    // (61 % of all tests on the atrium).  The reference returns from the first leaf cell in which ANY triangle is hit
    // (TwoLevelGrid.h:528-540), so whatever this ray tested before the current cell was a miss, and testing it again gives
    // the same miss: the test is skipped when the triangle sits in a 16-entry direct-mapped table of tested triangles
    // (per lane, LDS [entry][lane]; entry = id & 15, tag = the id).  Results are unchanged by construction.
    __shared__ mail_t s_mailbox[TLG_MAIL][TLG_BLOCK];
    // the planes of the three axes for the walk with its own records (k_tlg_trace_records; at most TLG_AXIS_MAX of them, else a.toprec is null)
    // (dynamic: its size is the grid's -- 16 bytes per plane; the launch passes it, so that small grids leave the LDS to more waves)
    extern __shared__ float4 s_axis[];
    if (a.toprec) {
        const uint32_t planes = a.grid.resX + a.grid.resY + a.grid.resZ;
        for (uint32_t i = threadIdx.x; i < planes; i += TLG_BLOCK) s_axis[i] = a.axis_tab[i];
        __syncthreads();
    }
    const uint32_t w = blockIdx.x * TLG_BLOCK + threadIdx.x;
    if (w >= a.total_work) return;
#pragma unroll
    for (int k = 0; k < (int)TLG_MAIL; k++) s_mailbox[k][threadIdx.x] = (mail_t)~0u;
    f3 o, d;
    // work item -> pixel: this rank's 64x64 tiles, inside a tile 8x8 blocks (one wave = one block), as in trace_kernels.hip; evaluated
    // here for the ray and again for the store (two registers that would live through the whole walk otherwise)
    auto locate = [&](uint32_t& x, uint32_t& y, uint32_t& slot) {
        const uint32_t ordinal = w >> 12, within = w & 4095u;
        const uint32_t tile = rt_tile_of(a.rank, ordinal, a.nranks);
        const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const uint32_t blk = within >> 6, lp = within & 63u;
        const uint32_t ly = ((blk >> 3) << 3) + (lp >> 3), lx = ((blk & 7u) << 3) + (lp & 7u);
        x = tx * RT_TILE + lx; y = ty * RT_TILE + ly;
        slot = (ordinal << 12) + ly * RT_TILE + lx;
    };
    if (PRIMARY) {
        uint32_t x, y, slot;
        locate(x, y, slot);
        if (x >= a.cam.resX || y >= a.cam.resY) {
            if (a.compact) {         // slot of a border tile outside the frame: defined as "no contact", as in the BVH kernels
                float4* z = reinterpret_cast<float4*>(a.out + slot);
                z[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                z[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            return;
        }
        generate_ray_xy(a.cam, x, y, o, d);
    } else {
        const float4* p = reinterpret_cast<const float4*>(a.rays + w);
        const float4 ro = p[1], rd = p[2];
        o = f3{ ro.x, ro.y, ro.z }; d = f3{ rd.x, rd.y, rd.z };
    }
    float best_t = 0.0f;                    // NO_CONTACT (CLStructs.h:124)
    uint32_t best_tri = 0xFFFFFFFFu;
    // the walk with its own records where the whole wave qualifies (wave-uniform: one of the two instances runs)
    const bool ray_ok = a.toprec != nullptr && tlg_moderate(o.x, d.x) && tlg_moderate(o.y, d.y) && tlg_moderate(o.z, d.z);
    if (__all(ray_ok)) {
        // ... and the packet walk where, besides, the wave's rays have the same direction signs (all but the blocks on the image's axes)
        const unsigned long long every = __ballot(1), bx = __ballot(d.x > 0.0f), by = __ballot(d.y > 0.0f), bz = __ballot(d.z > 0.0f);
        if ((bx == 0ull || bx == every) && (by == 0ull || by == every) && (bz == 0ull || bz == every) && (!PRIMARY || (a.tricam != nullptr && a.cam_off != 0u))) {
            switch ((bx == 0ull ? 1u : 0u) | (by == 0ull ? 2u : 0u) | (bz == 0ull ? 4u : 0u)) {          // (a scalar value: one instance per direction octant)
            case 0u: tlg_walk<PRIMARY, 2, 0>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 1u: tlg_walk<PRIMARY, 2, 1>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 2u: tlg_walk<PRIMARY, 2, 2>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 3u: tlg_walk<PRIMARY, 2, 3>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 4u: tlg_walk<PRIMARY, 2, 4>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 5u: tlg_walk<PRIMARY, 2, 5>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            case 6u: tlg_walk<PRIMARY, 2, 6>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            default: tlg_walk<PRIMARY, 2, 7>(a, o, d, s_mailbox, s_axis, best_t, best_tri); break;
            }
        }
        else tlg_walk<PRIMARY, 1>(a, o, d, s_mailbox, s_axis, best_t, best_tri);
    }
    else tlg_walk<PRIMARY, 0>(a, o, d, s_mailbox, s_axis, best_t, best_tri);

    // (what the store needs of the arguments -- camera, tiles, output, normals -- is read again from the kernel-argument segment here: kept in
    //  scalar registers through the walk it costs the walk's own scalar values their registers)
    const TlgTraceArgs& A = *tlg_args_again();
    float4 nt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t material = 0;
    if (best_tri != 0xFFFFFFFFu) {
        const float4 n = A.nrm[best_tri];
        nt = make_float4(n.x, n.y, n.z, best_t);
        material = __float_as_uint(A.v0[best_tri].w);
    }
    uint32_t pixel = w, out_index = w;
    if (PRIMARY) {
        const uint32_t ordinal = w >> 12, within = w & 4095u;
        const uint32_t tile = rt_tile_of(A.rank, ordinal, A.nranks);
        const uint32_t ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
        const uint32_t blk = within >> 6, lp = within & 63u;
        const uint32_t ly = ((blk >> 3) << 3) + (lp >> 3), lx = ((blk & 7u) << 3) + (lp & 7u);
        pixel = (ty * RT_TILE + ly) * A.cam.resX + tx * RT_TILE + lx;
        out_index = A.compact ? (ordinal << 12) + ly * RT_TILE + lx : pixel;
    }
    const uint4 head = make_uint4(pixel, material, 0u, 0u);
    float4* dst = reinterpret_cast<float4*>(A.out + out_index);
    store16_stream(dst, *reinterpret_cast<const float4*>(&head));
    store16_stream(dst + 1, nt);
    // a fused frame: what k_shade computes from the contact just written, at the pixel's own index (shade_kernels.hip; a miss is zeros)
    if (PRIMARY && A.fused_rgba) {
        float4 col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (best_tri != 0xFFFFFFFFu && best_t > 0.0f) {
            rt_contact c;
            c.pixelIndex = pixel; c.materialIndex = material; c.pad[0] = 0; c.pad[1] = 0;
            c.normalAndDistance[0] = nt.x; c.normalAndDistance[1] = nt.y; c.normalAndDistance[2] = nt.z; c.normalAndDistance[3] = nt.w;
            col = shade_one(A.fused_scene, A.cam, c);
        }
        store16_stream(A.fused_rgba + out_index, col);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 blocks_for(uint32_t n) { return dim3((n + TLG_BLOCK - 1) / TLG_BLOCK); }

void launch_tlg_count_top(hipStream_t s, const rt_grid_data& g, const DeviceScene& sc, uint32_t* counters)
{
    hipLaunchKernelGGL(k_tlg_count_top, blocks_for(sc.n_tris), dim3(TLG_BLOCK), 0, s, g, sc.v0, sc.v1, sc.v2, sc.n_tris, counters);
}
void launch_tlg_emit_top(hipStream_t s, const rt_grid_data& g, const DeviceScene& sc, const uint32_t* scan, const uint32_t* counters, uint2* pairs,
                         uint32_t pairs_cap)
{
    hipLaunchKernelGGL(k_tlg_emit_top, blocks_for(sc.n_tris), dim3(TLG_BLOCK), 0, s, g, sc.v0, sc.v1, sc.v2, sc.n_tris, scan, counters, pairs, pairs_cap);
}
// count_dev != null: `count` is the capacity the grid covers and the real count is read on the device
void launch_segment_bounds(hipStream_t s, const uint2* sorted, uint32_t count, const uint32_t* count_dev, uint2* ranges, uint32_t ranges_cap)
{
    if (count) hipLaunchKernelGGL(k_segment_bounds, blocks_for(count), dim3(TLG_BLOCK), 0, s, sorted, count, count_dev, ranges, ranges_cap);
}
void launch_tlg_cell_res(hipStream_t s, const rt_grid_data& g, const uint2* ranges, uint32_t cells, rt_top_cell* top, uint32_t* leaves)
{
    hipLaunchKernelGGL(k_tlg_cell_res, blocks_for(cells), dim3(TLG_BLOCK), 0, s, g, ranges, cells, top, leaves);
}
void launch_tlg_first_leaf(hipStream_t s, rt_top_cell* top, const uint32_t* scan, uint32_t cells)
{
    hipLaunchKernelGGL(k_tlg_first_leaf, blocks_for(cells), dim3(TLG_BLOCK), 0, s, top, scan, cells);
}
void launch_tlg_count_leaf(hipStream_t s, const rt_grid_data& g, const uint2* pairs, uint32_t pairs_count, const uint32_t* count_dev,
                           const rt_top_cell* top, const DeviceScene& sc, uint32_t* counters)
{
    if (pairs_count)
        hipLaunchKernelGGL(k_tlg_count_leaf, blocks_for(pairs_count), dim3(TLG_BLOCK), 0, s, g, pairs, pairs_count, count_dev, pairs_count, top,
                           sc.v0, sc.v1, sc.v2, counters);
}
void launch_tlg_emit_leaf(hipStream_t s, const rt_grid_data& g, const uint2* pairs, uint32_t pairs_count, const uint32_t* count_dev,
                          const rt_top_cell* top, const DeviceScene& sc, const uint32_t* scan, const uint32_t* counters, uint2* leaf_pairs,
                          uint32_t leaf_cap, uint32_t* written_total)
{
    if (pairs_count)
        hipLaunchKernelGGL(k_tlg_emit_leaf, blocks_for(pairs_count), dim3(TLG_BLOCK), 0, s, g, pairs, pairs_count, top, sc.v0, sc.v1, sc.v2,
                           scan, counters, leaf_pairs, written_total, count_dev, leaf_cap);
}

// the camera-relative triangle records (TriCam) from the scene's table: the operations of the test above that involve the origin
// and the triangle only, in its order
__global__ __launch_bounds__(TLG_BLOCK) void k_tlg_prepare_triangles(const float4* __restrict__ v0, const float4* __restrict__ e1, const float4* __restrict__ e2,
                                                                     TriCam* __restrict__ out, uint32_t n, float ox, float oy, float oz,
                                                                     const float4* __restrict__ axis_tab, TlgCamPlane* __restrict__ axis_cam, uint32_t res_x, uint32_t res_y, uint32_t planes)
{
    const uint32_t i = blockIdx.x * TLG_BLOCK + threadIdx.x;
    // the planes relative to this origin (the packet walk): near - o / far - o of dda_axis_setup and the operands of `inside` (dda_setup), once per camera
    if (axis_cam && i < planes) {
        const float4 pl = axis_tab[i];
        const float o = i < res_x ? ox : (i < res_x + res_y ? oy : oz);
        TlgCamPlane c;
        c.lo = pl.x; c.rcp_w = pl.z; c.w = pl.w;
        c.lo_o = pl.x - o; c.hi_o = pl.y - o; c.lo_o_again = c.lo_o;
        c.inside = (pl.x <= o) && (o <= pl.y) ? 1u : 0u;
        c.pad = 0u;
        axis_cam[i] = c;
    }
    if (i >= n) return;
    const f3 vert0 = f3_from4(v0[i]), edge1 = f3_from4(e1[i]), edge2 = f3_from4(e2[i]);
    const f3 tvec = f3_sub(f3{ ox, oy, oz }, vert0);
    const f3 qvec = f3_cross(tvec, edge1);
    TriCam c;
    c.e1[0] = edge1.x; c.e1[1] = edge1.y; c.e1[2] = edge1.z;
    c.e2[0] = edge2.x; c.e2[1] = edge2.y; c.e2[2] = edge2.z;
    c.tvec[0] = tvec.x; c.tvec[1] = tvec.y; c.tvec[2] = tvec.z;
    c.qvec[0] = qvec.x; c.qvec[1] = qvec.y; c.qvec[2] = qvec.z;
    c.e2_dot_qvec = f3_dot(edge2, qvec);
    c.pos = 0u; c.pad[0] = c.pad[1] = 0u;
    out[i] = c;
}

static uint32_t tlg_axis_bytes(const TlgTraceArgs& a) { return a.toprec ? (a.grid.resX + a.grid.resY + a.grid.resZ) * (uint32_t)sizeof(float4) : 0u; }
static TlgTraceArgs tlg_args(const TlgLaunch& t)
{
    TlgTraceArgs a{};
    a.grid = t.grid; a.top = t.top; a.leaf_ranges = t.leaf_ranges; a.leaf_pairs = t.leaf_pairs;
    a.v0 = t.scene->v0; a.e1 = t.scene->e1; a.e2 = t.scene->e2; a.nrm = t.scene->nrm;
    a.toprec = t.toprec; a.axis_tab = t.axis_tab; a.leafrec = t.leafrec;
    return a;
}

void launch_tlg_trace_primary(hipStream_t s, const TlgLaunch& t, const rt_camera& cam, uint32_t rank, uint32_t nranks, bool compact,
                              rt_contact* out, float4* fused_rgba)
{
    TlgTraceArgs a = tlg_args(t);
    a.cam = cam; a.rank = rank; a.nranks = nranks; a.compact = compact ? 1u : 0u;
    a.tiles_x = rt_tiles_x(cam.resX);
    const uint32_t tiles = rt_tiles_x(cam.resX) * rt_tiles_y(cam.resY);
    a.total_work = rt_owned_tiles(tiles, rank, nranks) * RT_TILE * RT_TILE;
    a.out = out;
    a.tricam = t.tricam;
    if (fused_rgba && !compact) {
        a.fused_rgba = fused_rgba;
        a.fused_scene.scene = t.scene->bytes; a.fused_scene.n_lights = t.scene->n_lights; a.fused_scene.n_spheres = t.scene->n_spheres;
        a.fused_scene.n_materials = t.scene->n_materials;
    }
    const uint32_t planes = t.grid.resX + t.grid.resY + t.grid.resZ;
    const bool cam_planes = t.tricam && t.axis_cam && t.axis_tab && t.toprec;
    a.cam_off = cam_planes ? t.cam_off : 0u;
    if (t.tricam && t.prepare_tricam && t.scene->n_tris)
        hipLaunchKernelGGL(k_tlg_prepare_triangles, blocks_for(t.scene->n_tris > planes ? t.scene->n_tris : planes), dim3(TLG_BLOCK), 0, s, t.scene->v0, t.scene->e1, t.scene->e2,
                           const_cast<TriCam*>(t.tricam), t.scene->n_tris, cam.viewTransform[3], cam.viewTransform[7], cam.viewTransform[11],
                           t.axis_tab, cam_planes ? t.axis_cam : nullptr, t.grid.resX, t.grid.resY, planes);
#ifdef RTMI_TLG_DIAG
    static unsigned long long zero[96] = { 0 };
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tlg_diag), zero, sizeof zero, 0, hipMemcpyHostToDevice, s);
#endif
    if (a.total_work) hipLaunchKernelGGL(k_tlg_trace<1>, blocks_for(a.total_work), dim3(TLG_BLOCK), tlg_axis_bytes(a), s, a);
#ifdef RTMI_TLG_DIAG
    if (getenv("RTMI_TLG_DIAG_PRINT")) {
        unsigned long long h[96];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tlg_diag), sizeof h);
        const char* nm[5] = { "top cells", "non-empty top cells", "leaf cells", "range entries", "triangle tests" };
        for (int k = 0; k < 5; k++)
            fprintf(stderr, "tlg diag %-20s wave iterations %12llu  lane iterations %14llu  lanes per wave iteration %.1f\n", nm[k], h[2 * k], h[2 * k + 1],
                    h[2 * k] ? (double)h[2 * k + 1] / (double)h[2 * k] : 0.0);
        const char* nm2[7] = { "uniform top cell", "uniform top cell, 64 lanes", "uniform leaf cell", "sign-uniform waves", "waves", "uniform triangle entry", "entries" };
        for (int k = 0; k < 7; k++)
            fprintf(stderr, "tlg diag %-28s wave iterations %12llu  lane iterations %14llu\n", nm2[k], h[80 + 2 * k], h[81 + 2 * k]);
        fprintf(stderr, "tlg diag leaf range sizes (lane visits):");
        for (int k = 0; k < 64; k++) fprintf(stderr, " %llu", h[16 + k]);
        fprintf(stderr, "\n");
    }
#endif
}

void launch_tlg_trace_rays(hipStream_t s, const TlgLaunch& t, const rt_ray* rays, uint32_t count, rt_contact* out)
{
    TlgTraceArgs a = tlg_args(t);
    a.rays = rays; a.total_work = count; a.out = out;
    if (count) hipLaunchKernelGGL(k_tlg_trace<0>, blocks_for(count), dim3(TLG_BLOCK), tlg_axis_bytes(a), s, a);
}

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_tlg()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_tlg_count_top)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
