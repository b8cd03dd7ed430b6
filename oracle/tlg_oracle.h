/*
 * tlg_oracle.h -- CPU restatement of the reference's Two-Level-Grid build and traversal
 * (BASELINE.json configs[3]).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as rt_oracle.h): tests/, smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: pinned on the reference's own kernels run on an MI355X for the stages that stay in range on a consistent
 * grid (see rt_oracle.h; tests/golden/reference_cl.json): prepareDataKernel (per-triangle cell counts),
 * countLeavesAndFillCellKernel (leaf resolutions with the DEVICE's pow: equal to the pow_third defined below on every cell of
 * all five bundled models) and generateContacts2Kernel (pixelIndex / materialIndex / t bit for bit with
 * orc_set_device_arith(1)).  The remaining build kernels index out of range as written (below) and are not run.
 * Every function cites the reference lines it follows.  Where the reference is undefined
 * or broken the FIXED reading of SURVEY.md section 8(c) is used and said in place:
 *   - all n triangles are counted / emitted (the kernels skip the last one, TwoLevelGridKernels.cl:48-49,63-64);
 *   - cell ranges are extracted over the whole sorted array (the kernel only works in work-group 0, :81,93),
 *     the last range closes at `count` (not count-1, :114,220), UINT_MAX holes never become a cell;
 *   - leaf ranges start zeroed (the reference never initialises that buffer, TwoLevelGridManager.cpp:434-437);
 *   - SAT-accepted leaf cells beyond the number the bounding-box pass counted are dropped (the reference would
 *     overwrite its neighbour's slots), and reported in orc_tlg_info.dropped_leaf_pairs;
 *   - pow(x, oneThird) (TwoLevelGrid.h:197, TwoLevelGridManager.cpp:578; OpenCL leaves pow to 16 ulp and the host
 *     branch to the C library) is DEFINED as the fp32 rounding of x^0.3333333432674408 computed by the
 *     fixed double-precision sequence of orc_pow_third, which the product restates operation for operation;
 *   - float -> uint conversions of NaN / negative / too large values give 0 (C leaves them undefined);
 *   - a DDA step whose `next` values are all NaN ends the ray with NO_CONTACT (the reference's
 *     `while (minimal != next[axis]) axis++` would run off the array, TwoLevelGrid.h:511-513,655-657).
 */
#ifndef TLG_ORACLE_H
#define TLG_ORACLE_H

#include "rt_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TwoLevelGridData.h:52-64 */
typedef struct {
    uint32_t resX, resY, resZ;
    float stepX, stepY, stepZ;
    float leafDensity;
    float padTo64;
    orc_aabb box;
} orc_grid_data;                                                              /* 64 B */
/* TwoLevelGridData.h:41-47 */
typedef struct { uint32_t resX, resY, resZ, firstLeafIdx; } orc_top_cell;      /* 16 B */

typedef struct {
    uint32_t cells, pairs_count, pairs_pow2, leaf_cells_count, leaf_pairs_count, leaf_pairs_pow2;
    uint32_t written_leaf_pairs;     /* leaf pairs that passed the exact triangle/box test (the rest are holes) */
    uint32_t dropped_leaf_pairs;     /* see header comment */
    uint32_t nonempty_cells, flat_cells;  /* top cells with triangles / with triangles but a zero resolution axis */
} orc_tlg_info;

typedef struct {
    uint64_t top_visits;     /* top-level cells stepped through              */
    uint64_t leaf_visits;    /* leaf cells stepped through                   */
    uint64_t tri_tests;      /* triangle tests                               */
    uint64_t hits;
    uint64_t dropped_rays;   /* rays ended by the NaN rule of the header comment */
    double   sum_t;
    uint64_t repeat_tests;   /* tests of a triangle found in a 16-entry direct-mapped per-ray cache of tested triangles */
} orc_tlg_stats;

typedef struct orc_tlg orc_tlg;

float     orc_pow_third(float x);
/* AABBTriangleIntersect, AABB.h:477-540 (OpenCL branch: sign() of the OpenCL C spec) */
int       orc_aabb_triangle_intersect(orc_f4 boxcenter, orc_f4 boxhalfsize, orc_f4 v0, orc_f4 v1, orc_f4 v2);

/* TwoLevelGridManager::calculateGridData / getResolution, TwoLevelGridManager.cpp:566-600.  0 = ok, 1 = degenerate box */
int       orc_tlg_grid_data(const void* scene, float top_density, float leaf_density, orc_grid_data* out);

/* TwoLevelGridManager::initializeFrame + construct, TwoLevelGridManager.cpp:160-469 */
orc_tlg*  orc_tlg_build(const void* scene, float top_density, float leaf_density);
void      orc_tlg_free(orc_tlg* g);
void      orc_tlg_info_get(const orc_tlg* g, orc_tlg_info* out);
const orc_grid_data* orc_tlg_grid(const orc_tlg* g);
const orc_top_cell*  orc_tlg_top_cells(const orc_tlg* g);        /* [cells]            */
const orc_pair*      orc_tlg_top_pairs(const orc_tlg* g);        /* [pairs_pow2] sorted */
const orc_pair*      orc_tlg_cell_ranges(const orc_tlg* g);      /* [cells] (x=first, y=end) */
const orc_pair*      orc_tlg_leaf_pairs(const orc_tlg* g);       /* [leaf_pairs_pow2] sorted, holes = (UINT_MAX,UINT_MAX) at the end */
const orc_pair*      orc_tlg_leaf_ranges(const orc_tlg* g);      /* [leaf_cells_count] */

/* tlg_generate_contact, TwoLevelGrid.h:557-678 + processTopLevelCell :419-545 */
void      orc_tlg_traverse(const orc_tlg* g, const orc_ray* ray, orc_contact* out, orc_tlg_stats* st);
/* generateContactsKernel / generateContacts2Kernel, TwoLevelGridKernels.cl:228-270 */
void      orc_tlg_trace_primary(const orc_tlg* g, const orc_camera* cam, orc_contact* contacts, orc_tlg_stats* st, int nthreads);
void      orc_tlg_trace_rays(const orc_tlg* g, const orc_ray* rays, uint32_t count, orc_contact* contacts,
                             orc_tlg_stats* st, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
