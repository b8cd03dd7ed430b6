/*
 * tlg_oracle.c -- CPU restatement of the reference's Two-Level Grid (see tlg_oracle.h for status and the FIXED
 * readings).  TEST INFRASTRUCTURE ONLY.  Paths are relative to /root/reference/OpenCLRayTracer/.
 */
#include "tlg_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* CLPortability.h:147-148 */
#define RMIN(X, Y) (((X) < (Y)) ? (X) : (Y))
#define RMAX(X, Y) (((X) > (Y)) ? (X) : (Y))

_Static_assert(sizeof(orc_grid_data) == 64, "GridData");
_Static_assert(sizeof(orc_top_cell) == 16, "TopLevelCell");

struct orc_tlg {
    const char* scene;
    uint32_t n;                       /* triangles */
    orc_grid_data grid;
    orc_tlg_info info;
    orc_top_cell* top_cells;
    orc_pair* top_pairs;
    orc_pair* cell_ranges;
    orc_pair* leaf_pairs;
    orc_pair* leaf_ranges;
    /* triangle cache (vertices + material), so the hot loops do not re-walk the scene buffer */
    orc_f4* tv;                       /* [3n] */
    uint32_t* tmat;                   /* [n]  */
};

/* float -> uint with the saturating reading of the header comment */
static uint32_t f2u(float x) { return (x >= 1.0f && x < 4294967296.0f) ? (uint32_t)x : 0u; }

/* largestPowerOfTwo(x) << 1, RTKernelUtils.h:160-164 as used by TwoLevelGridManager.cpp:165,167,239,398 */
static uint32_t pow2_above(uint32_t x)
{
    if (x == 0) return 1;
    uint32_t p = 1;
    while ((uint64_t)p * 2 <= x) p *= 2;
    return p * 2;
}

/* ------------------------------------------------------------------ */
/* pow(x, oneThird) as defined in tlg_oracle.h: plain IEEE double +,-,*,/ only, so the product can restate it
 * operation for operation on the device.  x^y with y = fp32(1/3) = 1/3 + E:  cbrt(x) * exp(E ln x).            */
/* ------------------------------------------------------------------ */
float orc_pow_third(float xf)
{
    if (xf != xf) return xf;
    if (xf == 0.0f) return 0.0f;
    if (xf < 0.0f) return NAN;
    if (xf > FLT_MAX) return xf;
    double x = (double)xf;                       /* always a normal double */
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    uint64_t mb = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m; memcpy(&m, &mb, 8);                /* m in [1,2), x = m * 2^e */
    /* ln m = 2 atanh(s), s = (m-1)/(m+1) <= 1/3 */
    double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 27.0;
    for (int k = 25; k >= 1; k -= 2) p = p * s2 + 1.0 / (double)k;       /* 1 + s2/3 + ... + s2^13/27 (Horner) */
    double lnx = (double)e * 0x1.62e42fefa39efp-1 + 2.0 * (s * p);
    /* cbrt: e = 3q + r, a = m * 2^r in [1,8), Halley iterations from 1.5 */
    int q = (e >= 0) ? e / 3 : -((-e + 2) / 3);
    int r = e - 3 * q;
    double a = m * (double)(1 << r);
    double y = 1.5;
    for (int i = 0; i < 7; i++) { double y3 = y * y * y; y = y * (y3 + 2.0 * a) / (2.0 * y3 + a); }
    uint64_t sb = (uint64_t)(q + 1023) << 52;
    double scale; memcpy(&scale, &sb, 8);
    double z = 0x1.5555556000000p-27 * lnx;      /* E = fp32(1/3) - 1/3 */
    double res = (y * scale) * (1.0 + z + 0.5 * z * z);
    return (float)res;
}

/* ------------------------------------------------------------------ */
/* AABB.h                                                              */
/* ------------------------------------------------------------------ */

/* calculateTriangleAABB, AABB.h:61-79 */
static orc_aabb triangle_aabb(orc_f4 v1, orc_f4 v2, orc_f4 v3)
{
    orc_aabb r;
    orc_f4 lo = { RMIN(v1.x, v2.x), RMIN(v1.y, v2.y), RMIN(v1.z, v2.z), 0.0f };
    orc_f4 hi = { RMAX(v1.x, v2.x), RMAX(v1.y, v2.y), RMAX(v1.z, v2.z), 0.0f };
    lo.x = RMIN(lo.x, v3.x); lo.y = RMIN(lo.y, v3.y); lo.z = RMIN(lo.z, v3.z);
    hi.x = RMAX(hi.x, v3.x); hi.y = RMAX(hi.y, v3.y); hi.z = RMAX(hi.z, v3.z);
    r.bounds[0] = lo; r.bounds[1] = hi;
    uint32_t xFlat = (r.bounds[1].x - r.bounds[0].x) < FLT_EPSILON;
    uint32_t yFlat = (r.bounds[1].y - r.bounds[0].y) < FLT_EPSILON;
    uint32_t zFlat = (r.bounds[1].z - r.bounds[0].z) < FLT_EPSILON;
    r.bounds[1].x += (float)xFlat * FLT_EPSILON;
    r.bounds[1].y += (float)yFlat * FLT_EPSILON;
    r.bounds[1].z += (float)zFlat * FLT_EPSILON;
    r.bounds[0].x -= (float)xFlat * FLT_EPSILON;
    r.bounds[0].y -= (float)yFlat * FLT_EPSILON;
    r.bounds[0].z -= (float)zFlat * FLT_EPSILON;
    return r;
}

/* one axis of AABBOverlaps, AABB.h:219-226: "is minimum of the maximums smaller than maximum of the minimums" */
static int axis_overlaps(float alo, float ahi, float blo, float bhi)
{
    return !(RMIN(ahi, bhi) < RMAX(alo, blo));
}

/* sign() of OpenCL C (6.12.4): 1 / -1 / +-0 / 0 for NaN */
static float cl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }

extern int orc_device_arith;     /* rt_oracle.c: 1 = dot / cross as ROCm's OpenCL C built-ins (fused), the device branch */
static float dot3(orc_f4 a, orc_f4 b)
{
    if (orc_device_arith) return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
static orc_f4 sub3(orc_f4 a, orc_f4 b) { orc_f4 r = { a.x - b.x, a.y - b.y, a.z - b.z, 0.0f }; return r; }
static orc_f4 cross3(orc_f4 a, orc_f4 b)
{
    if (orc_device_arith) {
        orc_f4 d = { fmaf(a.y, b.z, b.y * -a.z), fmaf(a.z, b.x, b.z * -a.x), fmaf(a.x, b.y, b.x * -a.y), 0.0f };
        return d;
    }
    orc_f4 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f };
    return r;
}

/* planeBoxOverlap, AABB.h:455-475 */
static int plane_box_overlap(orc_f4 normal, orc_f4 vert, orc_f4 maxbox)
{
    orc_f4 vmin, vmax;
    vmin.x = -cl_sign(normal.x) * maxbox.x - vert.x;
    vmax.x = cl_sign(normal.x) * maxbox.x - vert.x;
    vmin.y = -cl_sign(normal.y) * maxbox.y - vert.y;
    vmax.y = cl_sign(normal.y) * maxbox.y - vert.y;
    vmin.z = -cl_sign(normal.z) * maxbox.z - vert.z;
    vmax.z = cl_sign(normal.z) * maxbox.z - vert.z;
    vmin.w = vmax.w = 0.0f;
    int yes = dot3(normal, vmin) <= 0.0f;
    yes = yes && dot3(normal, vmax) >= 0.0f;
    return yes;
}

/* the nine edge tests, AABB.h:389-453: p = a*u - b*v on two vertices, rad = fa*h1 + fb*h2 */
#define AXISTEST(pa, pb, rad_)                                        \
    do { float mn = RMIN(pa, pb), mx = RMAX(pa, pb); float rad = (rad_); \
         yes = yes && (mn <= rad && mx >= -rad); } while (0)

int orc_aabb_triangle_intersect(orc_f4 c, orc_f4 h, orc_f4 v0, orc_f4 v1, orc_f4 v2)
{
    int yes = 1;
    v0 = sub3(c, v0); v1 = sub3(c, v1); v2 = sub3(c, v2);                    /* :493-495 */
    {
        orc_f4 e = sub3(v1, v0);
        float fex = fabsf(e.x), fey = fabsf(e.y), fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v2.y - e.y * v2.z, fez * h.y + fey * h.z);       /* X01 */
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v2.x + e.x * v2.z, fez * h.x + fex * h.z);     /* Y02 */
        AXISTEST(e.y * v1.x - e.x * v1.y, e.y * v2.x - e.x * v2.y, fey * h.x + fex * h.y);       /* Z12 */
        e = sub3(v2, v1);
        fex = fabsf(e.x); fey = fabsf(e.y); fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v2.y - e.y * v2.z, fez * h.y + fey * h.z);       /* X01 */
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v2.x + e.x * v2.z, fez * h.x + fex * h.z);     /* Y02 */
        AXISTEST(e.y * v0.x - e.x * v0.y, e.y * v1.x - e.x * v1.y, fey * h.x + fex * h.y);       /* Z0  */
        e = sub3(v0, v2);
        fex = fabsf(e.x); fey = fabsf(e.y); fez = fabsf(e.z);
        AXISTEST(e.z * v0.y - e.y * v0.z, e.z * v1.y - e.y * v1.z, fez * h.y + fey * h.z);       /* X2  */
        AXISTEST(-e.z * v0.x + e.x * v0.z, -e.z * v1.x + e.x * v1.z, fez * h.x + fex * h.z);     /* Y1  */
        AXISTEST(e.y * v1.x - e.x * v1.y, e.y * v2.x - e.x * v2.y, fey * h.x + fex * h.y);       /* Z12 */
    }
    float mn, mx;
    mn = RMIN(v0.x, RMIN(v1.x, v2.x)); mx = RMAX(v0.x, RMAX(v1.x, v2.x));
    yes = yes && (mn <= h.x && mx >= -h.x);
    mn = RMIN(v0.y, RMIN(v1.y, v2.y)); mx = RMAX(v0.y, RMAX(v1.y, v2.y));
    yes = yes && (mn <= h.y && mx >= -h.y);
    mn = RMIN(v0.z, RMIN(v1.z, v2.z)); mx = RMAX(v0.z, RMAX(v1.z, v2.z));
    yes = yes && (mn <= h.z && mx >= -h.z);
    orc_f4 normal = cross3(sub3(v1, v0), sub3(v2, v1));
    yes = yes && plane_box_overlap(normal, v0, h);
    return yes;
}

/* ------------------------------------------------------------------ */
/* grid data: TwoLevelGridManager.cpp:566-600                          */
/* ------------------------------------------------------------------ */
static orc_aabb scene_bbox(const char* scene) { orc_aabb b; memcpy(&b, scene + 64, 32); return b; }   /* SceneBufferParser.h:55-66 */

int orc_tlg_grid_data(const void* scene_, float top_density, float leaf_density, orc_grid_data* g)
{
    const char* scene = (const char*)scene_;
    memset(g, 0, sizeof *g);
    g->box = scene_bbox(scene);
    float deltaX = g->box.bounds[1].x - g->box.bounds[0].x;
    float deltaY = g->box.bounds[1].y - g->box.bounds[0].y;
    float deltaZ = g->box.bounds[1].z - g->box.bounds[0].z;
    float volume = deltaX * deltaY * deltaZ;                                    /* boxVolume, AABB.h:243-249 */
    float prims = (float)orc_scene_triangle_count(scene);
    float a = orc_pow_third(top_density * prims / volume);
    g->resX = f2u(deltaX * a); g->resY = f2u(deltaY * a); g->resZ = f2u(deltaZ * a);   /* fillVector3: float -> uint */
    if (g->resX == 0 || g->resY == 0 || g->resZ == 0) return 1;
    g->stepX = (g->box.bounds[1].x - g->box.bounds[0].x) / (float)g->resX;
    g->stepY = (g->box.bounds[1].y - g->box.bounds[0].y) / (float)g->resY;
    g->stepZ = (g->box.bounds[1].z - g->box.bounds[0].z) / (float)g->resZ;
    g->leafDensity = leaf_density;
    return 0;
}

/* ------------------------------------------------------------------ */
/* build                                                               */
/* ------------------------------------------------------------------ */

/* the float cell span of a triangle's bounding box, shared by countOverlappingCells (TwoLevelGrid.h:86-99)
 * and writeOverlappingPairs (:138-163) */
static void top_span(const orc_grid_data* g, const orc_f4* v, float start[3], float end[3])
{
    float lo[3] = { RMIN(v[0].x, RMIN(v[1].x, v[2].x)), RMIN(v[0].y, RMIN(v[1].y, v[2].y)), RMIN(v[0].z, RMIN(v[1].z, v[2].z)) };
    float hi[3] = { RMAX(v[0].x, RMAX(v[1].x, v[2].x)), RMAX(v[0].y, RMAX(v[1].y, v[2].y)), RMAX(v[0].z, RMAX(v[1].z, v[2].z)) };
    float org[3] = { g->box.bounds[0].x, g->box.bounds[0].y, g->box.bounds[0].z };
    float step[3] = { g->stepX, g->stepY, g->stepZ };
    float last[3] = { (float)(g->resX - 1), (float)(g->resY - 1), (float)(g->resZ - 1) };
    for (int k = 0; k < 3; k++) {
        float s = floorf((lo[k] - org[k]) / step[k]);
        float e = floorf((hi[k] - org[k]) / step[k]);
        start[k] = RMIN(s, last[k]);
        end[k] = RMIN(e, last[k]);
    }
}

static uint32_t cell_index(uint32_t ix, uint32_t iy, uint32_t iz, uint32_t rx, uint32_t ry)   /* getCellIndex :55-59 */
{
    return iz * rx * ry + iy * rx + ix;
}

/* geometry of the leaf cells of one top-level cell (shared by countOverlappingLeafCells :262-293 and
 * writeOverlappingLeafPairs :338-375) */
typedef struct { float stepX, stepY, stepZ, baseX, baseY, baseZ; } leaf_geom;
static leaf_geom leaf_geometry(const orc_grid_data* g, const orc_top_cell* tc, uint32_t topIdx)
{
    leaf_geom L;
    L.stepX = g->stepX / (float)tc->resX;
    L.stepY = g->stepY / (float)tc->resY;
    L.stepZ = g->stepZ / (float)tc->resZ;
    uint32_t a = g->resX * g->resY;                                            /* getCellRefFromIndex :69-75 */
    uint32_t z = topIdx / a, b = topIdx - a * z;
    uint32_t cx = b % g->resX, cy = b / g->resX;
    L.baseX = g->box.bounds[0].x + (float)cx * g->stepX;
    L.baseY = g->box.bounds[0].y + (float)cy * g->stepY;
    L.baseZ = g->box.bounds[0].z + (float)z * g->stepZ;
    return L;
}

void orc_tlg_free(orc_tlg* g)
{
    if (!g) return;
    free(g->top_cells); free(g->top_pairs); free(g->cell_ranges); free(g->leaf_pairs); free(g->leaf_ranges);
    free(g->tv); free(g->tmat); free(g);
}

/* extractCellRangesKernel / extractLeafCellsKernel (TwoLevelGridKernels.cl:71-116,176-221), FIXED reading */
static void extract_ranges(const orc_pair* sorted, uint32_t count, orc_pair* ranges)
{
    for (uint32_t i = 0; i + 1 < count; i++) {
        uint32_t right = sorted[i + 1].key, me = sorted[i].key;
        if (right > me) {
            ranges[me].value = i + 1;                         /* .y */
            if (right != UINT32_MAX) ranges[right].key = i + 1;  /* .x */
        }
    }
    if (count && sorted[count - 1].key != UINT32_MAX) ranges[sorted[count - 1].key].value = count;
}

orc_tlg* orc_tlg_build(const void* scene_, float top_density, float leaf_density)
{
    const char* scene = (const char*)scene_;
    orc_tlg* G = (orc_tlg*)calloc(1, sizeof *G);
    G->scene = scene;
    if (orc_tlg_grid_data(scene, top_density, leaf_density, &G->grid)) { free(G); return NULL; }
    const orc_grid_data* g = &G->grid;
    uint32_t n = (uint32_t)orc_scene_triangle_count(scene);
    G->n = n;
    G->tv = (orc_f4*)malloc(sizeof(orc_f4) * 3 * (size_t)n);
    G->tmat = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    for (uint32_t i = 0; i < n; i++) { uint32_t ref[3]; orc_scene_triangle(scene, i, G->tv + 3 * (size_t)i, ref, &G->tmat[i]); }

    uint32_t cells = g->resX * g->resY * g->resZ;
    G->info.cells = cells;
    uint32_t nPow2 = pow2_above(n), cellsPow2 = pow2_above(cells);

    /* prepareDataKernel -> prepareGridData (TwoLevelGridKernels.cl:43-51, TwoLevelGrid.h:108-123), all n triangles */
    uint32_t* counters = (uint32_t*)calloc(nPow2 > cellsPow2 ? nPow2 : cellsPow2, 4);
    uint32_t* scan = (uint32_t*)calloc(nPow2 > cellsPow2 ? nPow2 : cellsPow2, 4);
    for (uint32_t i = 0; i < n; i++) {
        float s[3], e[3];
        top_span(g, G->tv + 3 * (size_t)i, s, e);
        float cx = (e[0] - s[0]) + 1.0f, cy = (e[1] - s[1]) + 1.0f, cz = (e[2] - s[2]) + 1.0f;
        counters[i] = f2u(cx * cy * cz);
    }
    orc_scan_inclusive_u32(counters, scan, nPow2);
    uint32_t pairsCount = scan[n - 1];                                              /* TwoLevelGridManager.cpp:237 */
    uint32_t pairsPow2 = pow2_above(pairsCount);
    G->info.pairs_count = pairsCount; G->info.pairs_pow2 = pairsPow2;
    G->top_pairs = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)pairsPow2);
    memset(G->top_pairs, 0xff, sizeof(orc_pair) * (size_t)pairsPow2);              /* :248-250 */

    /* writePairsKernel -> writePairs -> writeOverlappingPairs (TwoLevelGrid.h:138-188) */
    for (uint32_t i = 0; i < n; i++) {
        float s[3], e[3];
        top_span(g, G->tv + 3 * (size_t)i, s, e);
        uint32_t c0[3], c1[3], mx[3] = { g->resX - 1, g->resY - 1, g->resZ - 1 };
        for (int k = 0; k < 3; k++) {   /* MIN3(convert_uint3(FLOOR3(..)), maxGridIdx): top_span already took the float min */
            c0[k] = f2u(s[k]); if (c0[k] > mx[k]) c0[k] = mx[k];
            c1[k] = f2u(e[k]); if (c1[k] > mx[k]) c1[k] = mx[k];
        }
        uint32_t first = scan[i] - counters[i], w = 0;
        for (uint32_t z = c0[2]; z <= c1[2]; z++)
            for (uint32_t y = c0[1]; y <= c1[1]; y++)
                for (uint32_t x = c0[0]; x <= c1[0]; x++) {
                    if (w < counters[i]) {
                        G->top_pairs[first + w].key = cell_index(x, y, z, g->resX, g->resY);
                        G->top_pairs[first + w].value = i;
                    }
                    w++;
                }
    }
    orc_bitonic_sort(G->top_pairs, pairsPow2);                                      /* :266 */

    /* extractCellRangesKernel (:291-317) */
    G->cell_ranges = (orc_pair*)calloc(cells, sizeof(orc_pair));
    extract_ranges(G->top_pairs, pairsCount, G->cell_ranges);

    /* countLeavesAndFillCellKernel -> fillTopLevelCell -> calcLeafCellResolution (TwoLevelGrid.h:193-225) */
    G->top_cells = (orc_top_cell*)calloc(cells, sizeof(orc_top_cell));
    memset(counters, 0, 4 * (size_t)(nPow2 > cellsPow2 ? nPow2 : cellsPow2));
    memset(scan, 0, 4 * (size_t)(nPow2 > cellsPow2 ? nPow2 : cellsPow2));
    for (uint32_t c = 0; c < cells; c++) {
        uint32_t prims = G->cell_ranges[c].value - G->cell_ranges[c].key;
        float volume = g->stepX * g->stepY * g->stepZ;
        float a = orc_pow_third(g->leafDensity * (float)prims / volume);
        orc_top_cell* tc = &G->top_cells[c];
        tc->resX = f2u(floorf(g->stepX * a)); tc->resY = f2u(floorf(g->stepY * a)); tc->resZ = f2u(floorf(g->stepZ * a));
        counters[c] = tc->resX * tc->resY * tc->resZ;
        if (prims) { G->info.nonempty_cells++; if (!counters[c]) G->info.flat_cells++; }
    }
    orc_scan_inclusive_u32(counters, scan, cellsPow2);
    uint32_t leafCells = scan[cells - 1];                                           /* :345 */
    G->info.leaf_cells_count = leafCells;
    for (uint32_t c = 1; c < cells; c++) G->top_cells[c].firstLeafIdx = scan[c - 1];  /* updateTopLevelCellsWithLeafRange :129-136 */

    /* prepareGridDataForLeaves -> countLeafPairs -> countOverlappingLeafCells (TwoLevelGrid.h:262-323) */
    uint32_t* lcount = (uint32_t*)calloc(pairsPow2, 4);
    uint32_t* lscan = (uint32_t*)calloc(pairsPow2, 4);
    for (uint32_t p = 0; p < pairsCount; p++) {
        uint32_t topIdx = G->top_pairs[p].key, tri = G->top_pairs[p].value;
        const orc_top_cell* tc = &G->top_cells[topIdx];
        const orc_f4* v = G->tv + 3 * (size_t)tri;
        orc_aabb tb = triangle_aabb(v[0], v[1], v[2]);
        leaf_geom L = leaf_geometry(g, tc, topIdx);
        uint32_t cnt = 0;
        for (uint32_t z = 0; z < tc->resZ; z++)
            for (uint32_t y = 0; y < tc->resY; y++)
                for (uint32_t x = 0; x < tc->resX; x++) {
                    float lx = L.baseX + (float)x * L.stepX, ly = L.baseY + (float)y * L.stepY, lz = L.baseZ + (float)z * L.stepZ;
                    float hx = lx + L.stepX, hy = ly + L.stepY, hz = lz + L.stepZ;
                    cnt += (uint32_t)(axis_overlaps(lx, hx, tb.bounds[0].x, tb.bounds[1].x) &&
                                      axis_overlaps(ly, hy, tb.bounds[0].y, tb.bounds[1].y) &&
                                      axis_overlaps(lz, hz, tb.bounds[0].z, tb.bounds[1].z));
                }
        lcount[p] = cnt;
    }
    orc_scan_inclusive_u32(lcount, lscan, pairsPow2);
    uint32_t leafPairs = lscan[pairsPow2 - 1];                                      /* :393 */
    uint32_t leafPairsPow2 = pow2_above(leafPairs);
    G->info.leaf_pairs_count = leafPairs; G->info.leaf_pairs_pow2 = leafPairsPow2;
    G->leaf_pairs = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)leafPairsPow2);
    memset(G->leaf_pairs, 0xff, sizeof(orc_pair) * (size_t)leafPairsPow2);          /* :404-408 */

    /* writeLeafPairsKernel -> writeLeafPairs -> writeOverlappingLeafPairs (TwoLevelGrid.h:338-403) */
    for (uint32_t p = 0; p < pairsCount; p++) {
        uint32_t topIdx = G->top_pairs[p].key, tri = G->top_pairs[p].value;
        const orc_top_cell* tc = &G->top_cells[topIdx];
        const orc_f4* v = G->tv + 3 * (size_t)tri;
        leaf_geom L = leaf_geometry(g, tc, topIdx);
        orc_f4 half = { L.stepX * 0.5f, L.stepY * 0.5f, L.stepZ * 0.5f, 0.0f };
        uint32_t first = lscan[p] - lcount[p], w = 0;
        for (uint32_t z = 0; z < tc->resZ; z++)
            for (uint32_t y = 0; y < tc->resY; y++)
                for (uint32_t x = 0; x < tc->resX; x++) {
                    orc_f4 c = { L.baseX + (float)x * L.stepX + half.x, L.baseY + (float)y * L.stepY + half.y,
                                 L.baseZ + (float)z * L.stepZ + half.z, 0.0f };
                    if (orc_aabb_triangle_intersect(c, half, v[0], v[1], v[2])) {
                        if (w < lcount[p]) {
                            G->leaf_pairs[first + w].key = cell_index(x, y, z, tc->resX, tc->resY) + tc->firstLeafIdx;
                            G->leaf_pairs[first + w].value = tri;
                            G->info.written_leaf_pairs++;
                        } else G->info.dropped_leaf_pairs++;
                        w++;
                    }
                }
    }
    orc_bitonic_sort(G->leaf_pairs, leafPairsPow2);                                 /* :424 */
    G->leaf_ranges = (orc_pair*)calloc(leafCells ? leafCells : 1, sizeof(orc_pair));
    extract_ranges(G->leaf_pairs, leafPairs, G->leaf_ranges);                       /* :439-466 */
    free(counters); free(scan); free(lcount); free(lscan);
    return G;
}

void orc_tlg_info_get(const orc_tlg* g, orc_tlg_info* out) { *out = g->info; }
const orc_grid_data* orc_tlg_grid(const orc_tlg* g) { return &g->grid; }
const orc_top_cell* orc_tlg_top_cells(const orc_tlg* g) { return g->top_cells; }
const orc_pair* orc_tlg_top_pairs(const orc_tlg* g) { return g->top_pairs; }
const orc_pair* orc_tlg_cell_ranges(const orc_tlg* g) { return g->cell_ranges; }
const orc_pair* orc_tlg_leaf_pairs(const orc_tlg* g) { return g->leaf_pairs; }
const orc_pair* orc_tlg_leaf_ranges(const orc_tlg* g) { return g->leaf_ranges; }

/* ------------------------------------------------------------------ */
/* traversal                                                           */
/* ------------------------------------------------------------------ */

/* the DDA set-up shared by both levels: TwoLevelGrid.h:436-500 (leaf level) and :575-640 (top level) */
typedef struct { float next[3], dt[3]; int idx[3], step[3], stop[3]; float t0, tmaxmax; } dda_t;

static void dda_setup(const orc_ray* ray, const orc_aabb* box, const uint32_t res[3], dda_t* d)
{
    float tmin[3], tmax[3];
    float o[3] = { ray->origin.x, ray->origin.y, ray->origin.z };
    float dir[3] = { ray->direction.x, ray->direction.y, ray->direction.z };
    float lo[3] = { box->bounds[0].x, box->bounds[0].y, box->bounds[0].z };
    float hi[3] = { box->bounds[1].x, box->bounds[1].y, box->bounds[1].z };
    for (int k = 0; k < 3; k++) {
        float a = 1.0f / dir[k];     /* z uses the double literal 1.0 (:445,584): same value after rounding to float */
        tmin[k] = ((a >= 0.0f ? lo[k] : hi[k]) - o[k]) * a;
        tmax[k] = ((a >= 0.0f ? hi[k] : lo[k]) - o[k]) * a;
    }
    d->t0 = RMIN(tmin[0], RMIN(tmin[1], tmin[2]));
    d->tmaxmax = RMAX(tmax[0], RMAX(tmax[1], tmax[2]));
    float p[3];
    if (orc_point_inside(box, ray->origin)) { p[0] = o[0]; p[1] = o[1]; p[2] = o[2]; }
    else { p[0] = o[0] + dir[0] * d->t0; p[1] = o[1] + dir[1] * d->t0; p[2] = o[2] + dir[2] * d->t0; }
    for (int k = 0; k < 3; k++) {
        float f = (p[k] - lo[k]) * (float)res[k] / (hi[k] - lo[k]);
        float hi_c = (float)res[k] - 1;
        f = RMAX(f, 0.0f); f = RMIN(f, hi_c);                    /* clamp(v,l,h) = min(max(v,l),h), CLPortability.h:401 */
        d->idx[k] = (int)f;
        d->dt[k] = (tmax[k] - tmin[k]) / (float)res[k];
        if (dir[k] > 0.0f) {
            d->next[k] = tmin[k] + (float)(d->idx[k] + 1) * d->dt[k];
            d->step[k] = +1; d->stop[k] = (int)res[k];
        } else {
            d->next[k] = dir[k] == 0.0f ? FLT_MAX : tmin[k] + (float)(res[k] - (uint32_t)d->idx[k]) * d->dt[k];
            d->step[k] = -1; d->stop[k] = -1;
        }
    }
}

/* `minimal = min(next[x], min(next[y], next[z])); while (minimal != next[axis]) axis++` ; -1 = no axis (FIXED) */
static int dda_axis(const dda_t* d)
{
    float minimal = RMIN(d->next[0], RMIN(d->next[1], d->next[2]));
    for (int axis = 0; axis < 3; axis++) if (!(minimal != d->next[axis])) return axis;
    return -1;
}

/* processTopLevelCell, TwoLevelGrid.h:419-545.  returns 1 and fills out when a leaf cell produced a hit */
static int process_top_cell(const orc_tlg* G, const orc_ray* ray, const orc_top_cell* tc, const orc_aabb* cellBox,
                            orc_contact* out, orc_tlg_stats* st, uint32_t* mbox)
{
    uint32_t res[3] = { tc->resX, tc->resY, tc->resZ };
    dda_t d;
    dda_setup(ray, cellBox, res, &d);
    orc_f4 best = { 0.0f, 0.0f, 0.0f, FLT_MAX };
    uint32_t material = 0;
    int found = 0;
    for (;;) {
        int axis = dda_axis(&d);
        uint32_t leafIndex = cell_index((uint32_t)d.idx[0], (uint32_t)d.idx[1], (uint32_t)d.idx[2], tc->resX, tc->resY) + tc->firstLeafIdx;
        orc_pair range = G->leaf_ranges[leafIndex];
        if (st) st->leaf_visits++;
        for (uint32_t i = range.key; i < range.value; i++) {
            uint32_t tri = G->leaf_pairs[i].value;
            const orc_f4* v = G->tv + 3 * (size_t)tri;
            orc_f4 c = orc_triangle_intersect(v[0], v[1], v[2], ray->origin, ray->direction);
            if (st) { st->tri_tests++; if (mbox[tri & 15u] == tri) st->repeat_tests++; mbox[tri & 15u] = tri; }
            if (c.w > 0 && c.w < best.w) { found = 1; best = c; material = G->tmat[tri]; }
        }
        if (found) {
            memset(out, 0, sizeof *out);
            out->materialIndex = material;
            out->normal_t = best;
            return 1;
        }
        if (axis < 0) { if (st) st->dropped_rays++; return 0; }
        d.next[axis] += d.dt[axis];
        d.idx[axis] += d.step[axis];
        if (d.idx[axis] == d.stop[axis]) return 0;
    }
}

void orc_tlg_traverse(const orc_tlg* G, const orc_ray* ray, orc_contact* out, orc_tlg_stats* st)
{
    const orc_grid_data* g = &G->grid;
    memset(out, 0, sizeof *out);                                               /* NO_CONTACT, CLStructs.h:124 */
    uint32_t res[3] = { g->resX, g->resY, g->resZ };
    dda_t d;
    uint32_t mbox[16];
    memset(mbox, 0xff, sizeof mbox);
    dda_setup(ray, &g->box, res, &d);
    if (d.t0 > d.tmaxmax) return;                                             /* :590-591 */
    for (;;) {
        const orc_top_cell* tc = &G->top_cells[cell_index((uint32_t)d.idx[0], (uint32_t)d.idx[1], (uint32_t)d.idx[2], g->resX, g->resY)];
        int axis = dda_axis(&d);
        if (st) st->top_visits++;
        if (!(tc->resX == 0 || tc->resY == 0 || tc->resZ == 0)) {
            orc_aabb cellBox;
            cellBox.bounds[0].x = g->box.bounds[0].x + (float)d.idx[0] * g->stepX;
            cellBox.bounds[0].y = g->box.bounds[0].y + (float)d.idx[1] * g->stepY;
            cellBox.bounds[0].z = g->box.bounds[0].z + (float)d.idx[2] * g->stepZ;
            cellBox.bounds[1].x = cellBox.bounds[0].x + g->stepX;
            cellBox.bounds[1].y = cellBox.bounds[0].y + g->stepY;
            cellBox.bounds[1].z = cellBox.bounds[0].z + g->stepZ;
            cellBox.bounds[0].w = cellBox.bounds[1].w = 0.0f;
            orc_contact c;
            if (process_top_cell(G, ray, tc, &cellBox, &c, st, mbox)) {
                *out = c;
                if (st) { st->hits++; st->sum_t += (double)c.normal_t.w; }
                return;
            }
        }
        if (axis < 0) { if (st) st->dropped_rays++; return; }
        d.next[axis] += d.dt[axis];
        d.idx[axis] += d.step[axis];
        if (d.idx[axis] == d.stop[axis]) return;
    }
}

/* ---- frame drivers ---- */
typedef struct {
    const orc_tlg* G; const orc_camera* cam; const orc_ray* rays; uint32_t first, count;
    orc_contact* contacts; orc_tlg_stats st; int want_stats;
} tjob;

static void* tworker(void* arg)
{
    tjob* j = (tjob*)arg;
    for (uint32_t i = j->first; i < j->first + j->count; i++) {
        orc_ray r;
        if (j->cam) orc_generate_ray(j->cam, i, &r); else r = j->rays[i];
        orc_tlg_traverse(j->G, &r, &j->contacts[i], j->want_stats ? &j->st : NULL);
        j->contacts[i].pixelIndex = i;                                        /* TwoLevelGridKernels.cl:243,266 */
    }
    return NULL;
}

static void run(const orc_tlg* G, const orc_camera* cam, const orc_ray* rays, uint32_t total, orc_contact* contacts,
                orc_tlg_stats* st, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > total && total) nthreads = (int)total;
    /* interleaved chunks of 4096 rays would balance better; rows are coherent enough: contiguous chunks, 8 per thread */
    int chunks = nthreads * 8;
    if ((uint32_t)chunks > total) chunks = total ? (int)total : 1;
    tjob* jobs = (tjob*)calloc((size_t)chunks, sizeof(tjob));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    uint32_t per = (total + (uint32_t)chunks - 1) / (uint32_t)chunks;
    for (int c = 0; c < chunks; c++) {
        uint32_t first = (uint32_t)c * per;
        jobs[c].G = G; jobs[c].cam = cam; jobs[c].rays = rays; jobs[c].contacts = contacts; jobs[c].want_stats = st != NULL;
        jobs[c].first = first < total ? first : total;
        jobs[c].count = first < total ? (first + per <= total ? per : total - first) : 0;
    }
    /* simple static round-robin of chunks over threads */
    for (int base = 0; base < chunks; base += nthreads) {
        int k = chunks - base < nthreads ? chunks - base : nthreads;
        for (int t = 0; t < k; t++) pthread_create(&th[t], NULL, tworker, &jobs[base + t]);
        for (int t = 0; t < k; t++) pthread_join(th[t], NULL);
    }
    if (st) {
        memset(st, 0, sizeof *st);
        for (int c = 0; c < chunks; c++) {
            st->top_visits += jobs[c].st.top_visits; st->leaf_visits += jobs[c].st.leaf_visits;
            st->tri_tests += jobs[c].st.tri_tests; st->hits += jobs[c].st.hits;
            st->dropped_rays += jobs[c].st.dropped_rays; st->sum_t += jobs[c].st.sum_t;
            st->repeat_tests += jobs[c].st.repeat_tests;
        }
    }
    free(jobs); free(th);
}

void orc_tlg_trace_primary(const orc_tlg* G, const orc_camera* cam, orc_contact* contacts, orc_tlg_stats* st, int nthreads)
{
    run(G, cam, NULL, cam->resX * cam->resY, contacts, st, nthreads);
}

void orc_tlg_trace_rays(const orc_tlg* G, const orc_ray* rays, uint32_t count, orc_contact* contacts, orc_tlg_stats* st, int nthreads)
{
    run(G, NULL, rays, count, contacts, st, nthreads);
}
