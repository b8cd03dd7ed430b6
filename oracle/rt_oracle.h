/*
 * rt_oracle.h -- CPU restatement of the reference's BVH ray-intersection path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: pinned on the reference's own DEVICE code, run on an MI355X (round 2).
 *   - The HOST branch of the reference's dual-target headers (OpenCLRayTracer/Include/CLData/..., #ifdef _WIN32) needs Boost
 *     (<boost/utility/binary.hpp>, CLPortability.h:161); Boost is absent from this image and stand-in headers are not
 *     allowed, so that branch is unbuildable here.  The reference ships no golden vectors or runnable tests.
 *   - The DEVICE branch needs neither: `make -C oracle ref_cl` compiles the reference's four .cl files for gfx950 with ROCm
 *     clang where they lie (oracle/_ref/cl_*.co; -fms-extensions for their back-slash includes is the only extra switch), and
 *     oracle/ref_cl.py launches those kernels through the HIP module API.  tests/golden/reference_cl.json holds what they
 *     produced (generator: tests/golden/make_golden_reference_cl.py); tests/test_cpu_reference_pins.py checks this
 *     restatement against it anywhere, tests/test_gpu_reference_cl.py re-runs the kernels on the GPU box.
 *   - Bit for bit on all five bundled models: calculateMortonCodes (leaf nodes + Morton pairs), the B8/B4/B2 bitonic network,
 *     buildRadixTree (guard as written = LITERAL_ZERO stage; FIXED differs in exactly node n-2 and its children's parent
 *     links), group_prefixSum, and pixelIndex / materialIndex / t of generateContacts2 on the FIXED and the LITERAL_ZERO
 *     tree -- the last with orc_set_device_arith(1): the device branch's dot() / cross() are OpenCL built-ins, which ROCm's
 *     library fuses, and the restatement's default is the host branch's separate multiplies and adds (t then differs by
 *     <= 1.1e-5 relative, the hit mask on one edge-on pixel of Cube).  Normals: fast_normalize on the device, 3e-7.
 *   - NOT pinned: computeBoundingBoxes (the reference kernel races on the MI355X: 37-12 172 boxes too small, differently
 *     per run; FIXED = its merges once both children are final), the host-only shading (Shading.h), scene packing beyond
 *     the vendored OBJ parser (oracle/ref_tinyobj_dump.cpp).
 *
 * All structs are byte-identical to the reference PODs (sizes asserted in
 * rt_oracle.c).  cl_float3 is a 16-byte type on the host (CL/cl_platform.h),
 * hence f4 everywhere.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, w; } orc_f4;

/* AABB.h:47-52 */
typedef struct { orc_f4 bounds[2]; } orc_aabb;                       /* 32 B */
/* BVHData.h:70-74 ; data[0]=parent, inner data[1]/[2]=childA/B,
 * leaf data[1]=triangle, data[2]=submesh, data[3]=model; type in bounds[0].w */
typedef struct { uint32_t data[4]; orc_aabb box; } orc_bvh_node;     /* 48 B */
/* CLStructs.h:44-51 */
typedef struct { float FOVDistance; uint32_t resX, resY, supersampling; float m[12]; } orc_camera; /* 64 B */
/* CLStructs.h:82-87 */
typedef struct { uint32_t idx; uint32_t pad_[3]; orc_f4 origin; orc_f4 direction; } orc_ray;        /* 48 B */
/* CLStructs.h:112-118 */
typedef struct { uint32_t pixelIndex, materialIndex, pad[2]; orc_f4 normal_t; } orc_contact;       /* 32 B */
/* Sorting pairs: uint2, key = .x, value = .y (SortKernels.cl:4-8) */
typedef struct { uint32_t key, value; } orc_pair;

/* per-trace statistics (reference traversal semantics) */
typedef struct {
    uint64_t inner_visits;   /* I: inner nodes whose two children were tested */
    uint64_t leaf_tests;     /* L: leaf triangles tested                     */
    uint64_t hits;           /* contacts with t > 0                          */
    uint32_t max_stack;      /* deepest stack pointer reached (reference array is 32) */
    uint32_t pad_;
    double   sum_t;          /* sum of t over hits (fixture checksum)        */
} orc_stats;

enum { ORC_MODE_FIXED = 0, ORC_MODE_LITERAL_ZERO = 1 };

size_t   orc_sizeof(const char* name);
/* 0 (default): dot / cross of the headers' host branch; 1: ROCm's OpenCL C built-ins, as the reference's kernels compute
 * them on the MI355X (see rt_oracle.c).  Global, set before tracing. */
void     orc_set_device_arith(int on);

/* ---- camera (OpenCLRayTracerDemo.cpp:202-217, CLStructs.h:60-74, Transform.h:397-413) */
void     orc_demo_camera(const void* scene, uint32_t resX, uint32_t resY, orc_camera* out);

/* ---- ray generation (CLStructs.h:95-106) */
void     orc_generate_ray(const orc_camera* cam, uint32_t pixelIndex, orc_ray* out);
void     orc_generate_rays(const orc_camera* cam, uint32_t first, uint32_t count, orc_ray* rays);   /* pixels [first, first + count) */

/* ---- primitives */
float    orc_aabb_intersect(const orc_aabb* box, orc_f4 ro, orc_f4 rd);      /* AABB.h:126-142 */
int      orc_point_inside(const orc_aabb* box, orc_f4 p);                     /* AABB.h:178-184 */
orc_f4   orc_triangle_intersect(orc_f4 v0, orc_f4 v1, orc_f4 v2, orc_f4 o, orc_f4 d); /* Triangle.h:66-107 */

/* ---- scene buffer (SceneBufferParser.h) */
uint64_t orc_scene_triangle_count(const void* scene);
int      orc_scene_triangle(const void* scene, uint32_t globalTri, orc_f4 v[3], uint32_t ref[3], uint32_t* material);

/* ---- BVH build (BVH.h:49-334, BVHKernels.cl:46-79, BVHManager.cpp:122-228, BitonicSort.cpp:118-189) */
uint32_t orc_morton_buffer_items(uint32_t n);                                 /* BVHManager.cpp:128-131 */
void     orc_morton_leaves(const void* scene, orc_bvh_node* nodes, orc_pair* pairs); /* pairs[N] pre-filled by caller */
void     orc_bitonic_sort(orc_pair* pairs, uint32_t N);
void     orc_radix_tree(orc_bvh_node* nodes, const orc_pair* sorted, uint32_t n, int mode);
void     orc_refit(orc_bvh_node* nodes, uint32_t n, int mode);
/* whole pipeline: nodes[2n-1] (zero-filled here), pairs[N] */
int      orc_build_bvh(const void* scene, int mode, orc_bvh_node* nodes, orc_pair* pairs);

/* ---- traversal (BVH.h:344-408) */
void     orc_traverse(const orc_ray* ray, const orc_bvh_node* nodes, uint32_t root, const void* scene,
                      orc_contact* out, orc_stats* st);
void     orc_trace_primary(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                           orc_contact* contacts, orc_stats* st, int nthreads);              /* BVHKernels.cl:84-98 */
void     orc_trace_primary_range(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                           uint32_t first, uint32_t count, orc_contact* contacts, orc_stats* st, int nthreads);
/* same as orc_trace_primary, also returning the inner-node visits of every ray */
void     orc_trace_primary_visits(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                           orc_contact* contacts, uint32_t* inner_visits, orc_stats* st, int nthreads);
void     orc_trace_rays(const orc_ray* rays, uint32_t count, const orc_bvh_node* nodes, uint32_t n,
                        const void* scene, orc_contact* contacts, orc_stats* st, int nthreads); /* BVHKernels.cl:103-117 */
/* brute force over every triangle with the same triangle test (self-consistency check only) */
void     orc_brute_primary(const orc_camera* cam, const void* scene, orc_contact* contacts, int nthreads);

/* ---- shading (Shading.h:74-166; frame loop OpenCLRayTracerDemo.cpp:298-316) */
void     orc_shade_blinn_phong(const orc_camera* cam, const orc_contact* c, const void* scene, orc_f4* rgba);
void     orc_shade_frame(const orc_camera* cam, const orc_contact* contacts, uint32_t count,
                         const void* scene, orc_f4* rgba /* [resX*resY], zeroed here */, int nthreads);

/* ---- secondary rays (EXTENSION, reference TODO 2.2; see rt_oracle.c): kind 0 = shadow rays towards `target`, 1 = reflection */
uint32_t orc_generate_secondary_rays(const orc_camera* cam /* or NULL */, const orc_ray* rays_in /* used when cam == NULL */,
                                     const orc_contact* contacts, uint32_t count, int kind, orc_f4 target, float eps,
                                     orc_ray* out /* [count] */);

/* ---- primitives used by the builders */
void     orc_scan_inclusive_u32(const uint32_t* in, uint32_t* out, uint32_t n);  /* PrefixSumKernels.cl:28-106 semantics */

/* ---- structure checks (ideas from Include/Testing/BVHTest.h:52-113) */
int      orc_check_hierarchy(const orc_bvh_node* nodes, uint32_t n, uint32_t* depth_out);

#ifdef __cplusplus
}
#endif
#endif
