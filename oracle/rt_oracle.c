/*
 * rt_oracle.c -- CPU restatement of the reference's BVH ray-intersection path.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the parity status: pinned on the
 * reference's own OpenCL C kernels run on an MI355X, tests/golden/reference_cl.json;
 * the headers' host branch cannot be built in this image, Boost absent).  Each
 * function cites the reference lines it restates.  Paths are relative to
 * /root/reference/OpenCLRayTracer/ ; "CLData/" = Include/CLData/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (x86-64 SSE2 => plain IEEE
 * binary32 arithmetic, no fused multiply-add, no x87 excess precision).  The
 * reference's host branch (#ifdef _WIN32) is what is restated: min/max are the
 * ternary macros of CLData/CLPortability.h:147-148, normalize is
 * cpu_normalize (CLPortability.h:151-159), clz is software_clz (:163-173).
 */
#include "rt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* CLPortability.h:147-148 -- note the NaN behaviour of the ternaries */
#define RMIN(X, Y) (((X) < (Y)) ? (X) : (Y))
#define RMAX(X, Y) (((X) > (Y)) ? (X) : (Y))

_Static_assert(sizeof(orc_aabb) == 32, "AABB");
_Static_assert(sizeof(orc_bvh_node) == 48, "BVHNode");
_Static_assert(sizeof(orc_camera) == 64, "Camera");
_Static_assert(sizeof(orc_ray) == 48, "Ray");
_Static_assert(sizeof(orc_contact) == 32, "Contact");

/* ------------------------------------------------------------------ */
/* scene buffer format: CLData/SceneBufferParser.h:55-66,139-160       */
/* ------------------------------------------------------------------ */
enum {
    SCENE_HEADER_SIZE = 96,  /* 8 x u64 + AABB                    (:55-66)   */
    LIGHT_SIZE = 16,         /* Primitives/Light.h:43-46                     */
    SPHERE_SIZE = 16,        /* Primitives/Sphere.h:42-45                    */
    MATERIAL_SIZE = 96,      /* Primitives/Material.h:46-58                  */
    MODEL_HEADER_SIZE = 64,  /* 4 x u64 + AABB                    (:129-136) */
    MESH_HEADER_SIZE = 48,   /* 6 x u64                           (:141-149) */
    VERTEX_SIZE = 16,        /* VERTEX_TYPE = CL_FLOAT3 (16 B)    (:154-155) */
    INDEX_SIZE = 2           /* INDEX_TYPE = CL_USHORT            (:156-157) */
};

size_t orc_sizeof(const char* name)
{
    if (!strcmp(name, "AABB")) return sizeof(orc_aabb);
    if (!strcmp(name, "BVHNode")) return sizeof(orc_bvh_node);
    if (!strcmp(name, "Camera")) return sizeof(orc_camera);
    if (!strcmp(name, "Ray")) return sizeof(orc_ray);
    if (!strcmp(name, "Contact")) return sizeof(orc_contact);
    if (!strcmp(name, "SceneHeader")) return SCENE_HEADER_SIZE;
    if (!strcmp(name, "ModelHeader")) return MODEL_HEADER_SIZE;
    if (!strcmp(name, "MeshHeader")) return MESH_HEADER_SIZE;
    if (!strcmp(name, "Material")) return MATERIAL_SIZE;
    return 0;
}

static uint64_t rd_u64(const char* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static orc_f4   rd_f4(const char* p)  { orc_f4 v; memcpy(&v, p, 16); return v; }
static uint16_t rd_u16(const char* p) { uint16_t v; memcpy(&v, p, 2); return v; }

/* SceneHeader field accessors (SceneBufferParser.h:55-66) */
static uint64_t sh_lights(const char* s)    { return rd_u64(s + 16); }
static uint64_t sh_spheres(const char* s)   { return rd_u64(s + 24); }
static uint64_t sh_materials(const char* s) { return rd_u64(s + 32); }
static uint64_t sh_triangles(const char* s) { return rd_u64(s + 56); }
static orc_aabb sh_bbox(const char* s)      { orc_aabb b; memcpy(&b, s + 64, 32); return b; }

/* SceneBufferParser.h:69-74 */
static const char* lights_ptr(const char* s)    { return s + SCENE_HEADER_SIZE; }
static const char* spheres_ptr(const char* s)   { return lights_ptr(s) + sh_lights(s) * LIGHT_SIZE; }
static const char* materials_ptr(const char* s) { return spheres_ptr(s) + sh_spheres(s) * SPHERE_SIZE; }
static const char* models_ptr(const char* s)    { return materials_ptr(s) + sh_materials(s) * MATERIAL_SIZE; }

/* getModelAtIndex, SceneBufferParser.h:175-181: linear walk over ModelHeader.dataSize */
static const char* model_at(uint32_t index, const char* scene)
{
    const char* p = models_ptr(scene);
    for (uint32_t i = 0; i < index; i++) p += rd_u64(p); /* ModelHeader.dataSize */
    return p;
}
/* getMeshAtIndex, SceneBufferParser.h:189-195 */
static const char* mesh_at(uint32_t index, const char* model)
{
    const char* p = model + MODEL_HEADER_SIZE;
    for (uint32_t i = 0; i < index; i++) p += rd_u64(p); /* MeshHeader.dataSize */
    return p;
}
/* MeshHeader fields (:141-149) */
static uint64_t mesh_tris(const char* m)     { return rd_u64(m + 8); }
static uint64_t mesh_verts(const char* m)    { return rd_u64(m + 16); }
static uint64_t mesh_material(const char* m) { return rd_u64(m + 32); }
/* getVertexAt / getIndexAt, SceneBufferParser.h:203-217 (VERTEX_BASE/INDEX_BASE :162-163) */
static orc_f4 vertex_at(uint32_t index, const char* mesh)
{
    return rd_f4(mesh + MESH_HEADER_SIZE + (size_t)index * VERTEX_SIZE);
}
static uint16_t index_at(uint32_t index, const char* mesh)
{
    return rd_u16(mesh + MESH_HEADER_SIZE + mesh_verts(mesh) * VERTEX_SIZE + (size_t)index * INDEX_SIZE);
}

uint64_t orc_scene_triangle_count(const void* scene) { return sh_triangles((const char*)scene); }

/* getTriangleRefByIndex, SceneBufferParser.h:253-293 */
static void triangle_ref_by_index(const char* scene, uint32_t triangleIndex, uint32_t ref[3])
{
    int currentModelIndex = -1, currentSubmeshIndex = -1;
    int accumulated = 0;
    int flag = 0;
    const char* currentModel = NULL;
    while (!flag) {
        currentModelIndex++;
        currentModel = model_at((uint32_t)currentModelIndex, scene);
        accumulated += (int)rd_u64(currentModel + 16); /* ModelHeader.numberOfTriangles */
        flag = (uint32_t)accumulated > triangleIndex;
    }
    accumulated -= (int)rd_u64(currentModel + 16);
    flag = 0;
    const char* currentMesh = NULL;
    while (!flag) {
        currentSubmeshIndex++;
        currentMesh = mesh_at((uint32_t)currentSubmeshIndex, currentModel);
        accumulated += (int)mesh_tris(currentMesh);
        flag = (uint32_t)accumulated > triangleIndex;
    }
    accumulated -= (int)mesh_tris(currentMesh);
    ref[0] = (uint32_t)currentModelIndex;
    ref[1] = (uint32_t)currentSubmeshIndex;
    ref[2] = triangleIndex - (uint32_t)accumulated; /* max((uint)0, ...) is a no-op */
}

int orc_scene_triangle(const void* scene_, uint32_t globalTri, orc_f4 v[3], uint32_t ref[3], uint32_t* material)
{
    const char* scene = (const char*)scene_;
    if (globalTri >= sh_triangles(scene)) return 1;
    triangle_ref_by_index(scene, globalTri, ref);
    const char* mesh = mesh_at(ref[1], model_at(ref[0], scene));
    uint32_t base = ref[2] * 3;
    v[0] = vertex_at(index_at(base, mesh), mesh);
    v[1] = vertex_at(index_at(base + 1, mesh), mesh);
    v[2] = vertex_at(index_at(base + 2, mesh), mesh);
    if (material) *material = (uint32_t)mesh_material(mesh);
    return 0;
}

/* ------------------------------------------------------------------ */
/* vector helpers: CLPortability.h:203-279                             */
/* ------------------------------------------------------------------ */
static orc_f4 v_sub(orc_f4 a, orc_f4 b) { orc_f4 r = { a.x - b.x, a.y - b.y, a.z - b.z, 0.0f }; return r; }
static orc_f4 v_add(orc_f4 a, orc_f4 b) { orc_f4 r = { a.x + b.x, a.y + b.y, a.z + b.z, 0.0f }; return r; }
static orc_f4 v_scale(orc_f4 a, float s) { orc_f4 r = { a.x * s, a.y * s, a.z * s, 0.0f }; return r; }
/* Which definitions of dot() / cross() the restatement computes with.
 *   0 (default): the headers' HOST branch, CLPortability.h:266-278 -- separate multiplies and adds.
 *   1: the headers' DEVICE branch, where dot and cross are the OpenCL C built-ins (CLPortability.h:41-45 leaves them to
 *      the compiler's library).  ROCm 7.2's library (amdgcn/bitcode/opencl.bc, functions _Z3dotDv3_fS_ and _Z5crossDv3_fS_)
 *      defines them with fused multiply-adds, restated below; with this switch on, the oracle reproduces the t, hit mask and
 *      materialIndex of the reference's own generateContacts2 kernel running on the MI355X bit for bit
 *      (tests/test_gpu_reference_cl.py).  Test infrastructure only; the product restates the host branch. */
int orc_device_arith = 0;
void orc_set_device_arith(int on) { orc_device_arith = on; }
/* CLPortability.h:266-273 */
static orc_f4 v_cross(orc_f4 a, orc_f4 b)
{
    if (orc_device_arith) {
        orc_f4 d = { fmaf(a.y, b.z, b.y * -a.z), fmaf(a.z, b.x, b.z * -a.x), fmaf(a.x, b.y, b.x * -a.y), 0.0f };
        return d;
    }
    orc_f4 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f };
    return r;
}
/* CLPortability.h:275-278 */
static float v_dot(orc_f4 a, orc_f4 b)
{
    if (orc_device_arith) return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
/* cpu_normalize, CLPortability.h:151-159: l = 1/sqrt(|v|^2), then three multiplies */
static orc_f4 v_normalize(orc_f4 v)
{
    float l = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    orc_f4 r = { v.x * l, v.y * l, v.z * l, 0.0f };
    return r;
}
/* LENGTH3, CLPortability.h:280 */
static float v_length(orc_f4 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

/* software_clz(cl_uint), CLPortability.h:163-173 : clz(0) == 32 */
static uint32_t r_clz(uint32_t x)
{
    uint32_t mask = 0x80000000u, counter = 0;
    while ((x & mask) == 0 && counter < 32) { mask >>= 1; counter++; }
    return counter;
}

/* ------------------------------------------------------------------ */
/* camera + ray generation                                             */
/* ------------------------------------------------------------------ */

/* OpenCLRayTracerDemo.cpp:202-217 (camera placement), CLStructs.h:60-74
 * (PI, DEG2RAD, FOVDistFromAngle), RTFrame :275-276 -> normalizeQuaternion
 * (Transform.h:88-106) of zeroRotation() and setOrientationAndPos (:397-413). */
void orc_demo_camera(const void* scene, uint32_t resX, uint32_t resY, orc_camera* cam)
{
    const float PI = 3.14159265359f;
    const float DEG2RAD = PI / 180.0f;
    orc_aabb bb = sh_bbox((const char*)scene);
    memset(cam, 0, sizeof *cam);
    cam->resX = resX;
    cam->resY = resY;
    {   /* FOVDistFromAngle(90, winW, winH): a = resy*0.5f; a / tan(angle*0.5f*DEG2RAD) */
        float a = (float)resY * 0.5f;
        cam->FOVDistance = a / tanf(90.0f * 0.5f * DEG2RAD);
    }
    cam->supersampling = 1;
    float hx = (bb.bounds[1].x - bb.bounds[0].x) * 0.5f;
    float hy = (bb.bounds[1].y - bb.bounds[0].y) * 0.5f;
    float hz = (bb.bounds[1].z - bb.bounds[0].z) * 0.5f;
    float px = bb.bounds[0].x + hx, py = bb.bounds[0].y + hy, pz = bb.bounds[0].z - hz;
    /* quaternion (r,i,j,k) = (1,0,0,0); normalizeQuaternion: d = 1 -> 1/sqrt(1) = 1 */
    float r = 1.0f, i = 0.0f, j = 0.0f, k = 0.0f;
    float d = r * r + i * i + j * j + k * k;
    d = 1.0f / sqrtf(d);
    r *= d; i *= d; j *= d; k *= d;
    float* m = cam->m;
    m[0] = 1 - (2 * j * j + 2 * k * k); m[1] = 2 * i * j + 2 * k * r; m[2] = 2 * i * k - 2 * j * r; m[3] = px;
    m[4] = 2 * i * j - 2 * k * r; m[5] = 1 - (2 * i * i + 2 * k * k); m[6] = 2 * j * k + 2 * i * r; m[7] = py;
    m[8] = 2 * i * k + 2 * j * r; m[9] = 2 * j * k - 2 * i * r; m[10] = 1 - (2 * i * i + 2 * j * j); m[11] = pz;
}

/* generateRay, CLStructs.h:95-106; transformVectorByMatrix_const, Transform.h:282-295
 * (the translation column IS added to the direction); normalize = cpu_normalize;
 * origin = getTranslate_const (Transform.h:312-316). */
void orc_generate_ray(const orc_camera* cam, uint32_t pixelIndex, orc_ray* ray)
{
    orc_f4 d;
    d.x = -((float)(pixelIndex % cam->resX) - (float)cam->resX * 0.5f);
    d.y = (float)(pixelIndex / cam->resX) - (float)cam->resY * 0.5f;
    d.z = cam->FOVDistance;
    d.w = 0.0f;
    const float* m = cam->m;
    orc_f4 v;
    v.x = d.x * m[0] + d.y * m[1] + d.z * m[2] + m[3];
    v.y = d.x * m[4] + d.y * m[5] + d.z * m[6] + m[7];
    v.z = d.x * m[8] + d.y * m[9] + d.z * m[10] + m[11];
    v.w = 0.0f;
    memset(ray, 0, sizeof *ray);
    ray->direction = v_normalize(v);
    ray->origin.x = m[3]; ray->origin.y = m[7]; ray->origin.z = m[11];
    ray->idx = pixelIndex;
}

/* the whole primary ray buffer of a camera, pixels [first, first + count): what a host that feeds generateContacts(CLBuffer& rays, ..)
 * (BVHManager.cpp:292-329) with the camera's own rays would upload */
void orc_generate_rays(const orc_camera* cam, uint32_t first, uint32_t count, orc_ray* rays)
{
    for (uint32_t k = 0; k < count; k++) orc_generate_ray(cam, first + k, rays + k);
}

/* ------------------------------------------------------------------ */
/* ray / box, ray / triangle                                           */
/* ------------------------------------------------------------------ */

/* AABBIntersect, AABB.h:126-142 (INV_DIR_* :115-117 are re-evaluated at every use:
 * same correctly-rounded quotient each time).  Returns tmin*flag. */
float orc_aabb_intersect(const orc_aabb* aabb, orc_f4 ro, orc_f4 rd)
{
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;
    tmin  = (aabb->bounds[0.0f > ix].x - ro.x) * ix;
    tmax  = (aabb->bounds[1 - (0.0f > ix)].x - ro.x) * ix;
    tymin = (aabb->bounds[0.0f > iy].y - ro.y) * iy;
    tymax = (aabb->bounds[1 - (0.0f > iy)].y - ro.y) * iy;
    int flag = !((tmin > tymax) || (tymin > tmax));
    tmin = RMAX(tmin, tymin);
    tmax = RMIN(tmax, tymax);
    tzmin = (aabb->bounds[0.0f > iz].z - ro.z) * iz;
    tzmax = (aabb->bounds[1 - (0.0f > iz)].z - ro.z) * iz;
    flag &= !((tmin > tzmax) || (tzmin > tmax));
    tmin = RMAX(tmin, tzmin);
    tmax = RMIN(tmax, tzmax);
    (void)tmax;
    return tmin * (float)flag;
}

/* isPointInside, AABB.h:178-184; containedInRange, RTKernelUtils.h:52 */
int orc_point_inside(const orc_aabb* b, orc_f4 p)
{
    int xIn = (b->bounds[0].x <= p.x) && (p.x <= b->bounds[1].x);
    int yIn = (b->bounds[0].y <= p.y) && (p.y <= b->bounds[1].y);
    int zIn = (b->bounds[0].z <= p.z) && (p.z <= b->bounds[1].z);
    return xIn && yIn && zIn;
}

/* triangleIntersect, Triangle.h:66-107 */
orc_f4 orc_triangle_intersect(orc_f4 vert0, orc_f4 vert1, orc_f4 vert2, orc_f4 orig, orc_f4 dir)
{
    orc_f4 edge1 = v_sub(vert1, vert0);
    orc_f4 edge2 = v_sub(vert2, vert0);
    orc_f4 pvec = v_cross(dir, edge2);
    float det = v_dot(edge1, pvec);
    int flag = fabsf(det) >= FLT_EPSILON;
    float inv_det = 1.0f / det;
    orc_f4 tvec = v_sub(orig, vert0);
    float u = v_dot(tvec, pvec) * inv_det;
    flag &= (u < 1.0f);
    flag &= (u >= 0.0f);
    orc_f4 qvec = v_cross(tvec, edge1);
    float v = v_dot(dir, qvec) * inv_det;
    flag &= ((u + v) <= 1.0f);
    flag &= (v >= 0.0f);
    float t = v_dot(edge2, qvec) * inv_det * (float)flag;
    orc_f4 normal = v_normalize(v_cross(edge1, edge2));
    orc_f4 result = { normal.x, normal.y, normal.z, t };
    return result;
}

/* ------------------------------------------------------------------ */
/* BVH build                                                           */
/* ------------------------------------------------------------------ */

/* expandBits / morton3D, BVH.h:49-73 */
static uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static uint32_t morton3d(float x, float y, float z)
{
    x = RMIN(RMAX(x * 1024.0f, 0.0f), 1023.0f);
    y = RMIN(RMAX(y * 1024.0f, 0.0f), 1023.0f);
    z = RMIN(RMAX(z * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = expand_bits((uint32_t)x);
    uint32_t yy = expand_bits((uint32_t)y);
    uint32_t zz = expand_bits((uint32_t)z);
    return xx * 4 + yy * 2 + zz;
}

/* translateScale/normalizeScale, RTKernelUtils.h:63-81: value/oldLen, oldMin is NOT subtracted */
static float normalize_scale(float oldMin, float oldMax, float value)
{
    float oldLen = oldMax - oldMin;
    float percentage = value / oldLen;
    float newLen = 1.0f - 0.0f;
    return percentage * newLen + 0.0f;
}

/* calculateTriangleAABB, AABB.h:61-79 (MIN3/MAX3 = component-wise ternary min/max,
 * CLPortability.h:288-311; .w = 0) */
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

/* BVHManager.cpp:128-131: largestPowerOfTwo(n) (RTKernelUtils.h:160-164), doubled when < n */
uint32_t orc_morton_buffer_items(uint32_t n)
{
    uint32_t p = 1;
    while ((uint64_t)p * 2 <= n) p *= 2;   /* largest power of two <= n */
    if (p < n) p *= 2;
    return p;
}

/* kernel calculateMortonCodes (BVHKernels.cl:46-50) -> calculateMorton (BVH.h:83-130),
 * host branch: centroid per component (a+b+c)/3.0f.  pairs[] must hold N entries pre-filled
 * with (UINT_MAX,UINT_MAX) (BVHManager.cpp:136-139); leaf data[0] (parent) is left as found. */
void orc_morton_leaves(const void* scene_, orc_bvh_node* nodes, orc_pair* pairs)
{
    const char* scene = (const char*)scene_;
    uint32_t n = (uint32_t)sh_triangles(scene);
    orc_aabb sceneBB = sh_bbox(scene);
    for (uint32_t leaf = 0; leaf < n; leaf++) {
        uint32_t ref[3];
        triangle_ref_by_index(scene, leaf, ref);
        const char* mesh = mesh_at(ref[1], model_at(ref[0], scene));
        uint32_t base = ref[2] * 3;
        orc_f4 v1 = vertex_at(index_at(base, mesh), mesh);
        orc_f4 v2 = vertex_at(index_at(base + 1, mesh), mesh);
        orc_f4 v3 = vertex_at(index_at(base + 2, mesh), mesh);
        orc_bvh_node result;
        result.data[0] = nodes[leaf].data[0];
        result.box = triangle_aabb(v1, v2, v3);
        v1.x = (v1.x + v2.x + v3.x) / 3.0f;
        v1.y = (v1.y + v2.y + v3.y) / 3.0f;
        v1.z = (v1.z + v2.z + v3.z) / 3.0f;
        float nx = normalize_scale(sceneBB.bounds[0].x, sceneBB.bounds[1].x, v1.x);
        float ny = normalize_scale(sceneBB.bounds[0].y, sceneBB.bounds[1].y, v1.y);
        float nz = normalize_scale(sceneBB.bounds[0].z, sceneBB.bounds[1].z, v1.z);
        result.box.bounds[0].w = 1.0f;          /* type(result) = LEAF_NODE (BVHData.h:65) */
        result.data[1] = ref[2];                /* triangleIndex */
        result.data[2] = ref[1];                /* submeshIndex  */
        result.data[3] = ref[0];                /* modelIndex    */
        nodes[leaf] = result;
        pairs[leaf].key = morton3d(nx, ny, nz);
        pairs[leaf].value = leaf;
    }
}

/* BitonicSort::sort (BitonicSort.cpp:118-189) launching ParallelBitonic_B2/B4/B8
 * (SortKernels.cl:266-345).  B4/B8 fuse consecutive `inc` levels with the same comparators,
 * so the network is: for length, for inc: compare-exchange (i, i+inc) for every i with
 * (i & inc) == 0; reverse = ((dir & i) == 0) with dir = 2*length; swap = reverse ^ (a.key < b.key). */
void orc_bitonic_sort(orc_pair* d, uint32_t N)
{
    for (uint32_t length = 1; length < N; length <<= 1) {
        uint32_t dir = length << 1;
        for (uint32_t inc = length; inc > 0; inc >>= 1) {
            for (uint32_t i = 0; i < N; i++) {
                if (i & inc) continue;
                int reverse = ((dir & i) == 0);
                int swap = reverse ^ (d[i].key < d[i + inc].key);
                if (swap) { orc_pair t = d[i]; d[i] = d[i + inc]; d[i + inc] = t; }
            }
        }
    }
}

/* findSplit, BVH.h:138-181 */
static int find_split(const orc_pair* list, int first, int last)
{
    uint32_t firstCode = list[first].key;
    uint32_t lastCode = list[last].key;
    if (firstCode == lastCode) return first;
    int commonPrefix = (int)r_clz(firstCode ^ lastCode);
    int split = first;
    int step = last - first;
    do {
        step = (step + 1) >> 1;
        int newSplit = split + step;
        if (newSplit < last) {
            uint32_t splitCode = list[newSplit].key;
            int splitPrefix = (int)r_clz(firstCode ^ splitCode);
            if (splitPrefix > commonPrefix) split = newSplit;
        }
    } while (step > 1);
    return split;
}

/* determineRange, BVH.h:190-282 (NekoEngine duplicate-key variant) */
static void determine_range(const orc_pair* list, int index, uint32_t size, uint32_t* rx, uint32_t* ry)
{
    int lso = (int)size - 1;
    if (index == 0) { *rx = 0; *ry = (uint32_t)lso; return; }
    int dir, d_min;
    int initialindex = index;
    uint32_t minone = list[index - 1].key;
    uint32_t precis = list[index].key;
    uint32_t pluone = list[index + 1].key;
    if (minone == precis && pluone == precis) {
        while (index > 0 && index < lso) {
            index += 1;
            if (index >= lso || list[index].key != list[index + 1].key) break;
        }
        *rx = (uint32_t)initialindex; *ry = (uint32_t)index;
        return;
    } else {
        uint32_t lx = r_clz(precis ^ minone);
        uint32_t ly = r_clz(precis ^ pluone);
        if (lx > ly) { dir = -1; d_min = (int)ly; }
        else         { dir = 1;  d_min = (int)lx; }
    }
    int l_max = 2;
    int testindex = index + l_max * dir;
    while ((testindex <= lso && testindex >= 0) ? (r_clz(precis ^ list[testindex].key) > (uint32_t)d_min) : 0) {
        l_max = l_max << 1;
        testindex = index + l_max * dir;
    }
    int l = 0;
    for (int div = 2; l_max / div >= 1; div = div << 1) {
        int t = l_max / div;
        int newTest = index + (l + t) * dir;
        if (newTest <= lso && newTest >= 0) {
            int splitPrefix = (int)r_clz(precis ^ list[newTest].key);
            if (splitPrefix > d_min) l = l + t;
        }
    }
    int a = index, b = index + l * dir;
    *rx = (uint32_t)RMIN(a, b);
    *ry = (uint32_t)RMAX(a, b);
}

/* constructNode, BVH.h:291-321 */
static void construct_node(orc_bvh_node* nodes, const orc_pair* mc, uint32_t numLeaves, int idx, int literal)
{
    uint32_t rx, ry;
    determine_range(mc, idx, numLeaves, &rx, &ry);
    int split = find_split(mc, (int)rx, (int)ry);
    uint32_t internal = (uint32_t)idx + numLeaves;
    uint32_t first = RMIN(rx, ry), last = RMAX(rx, ry);
    int aIsLeaf = first == (uint32_t)split;
    int bIsLeaf = (last == (uint32_t)(split + 1));
    uint32_t child_A = mc[split].value * (uint32_t)aIsLeaf + ((uint32_t)split + numLeaves) * (uint32_t)!aIsLeaf;
    uint32_t child_B = mc[split + 1].value * (uint32_t)bIsLeaf + ((uint32_t)split + 1 + numLeaves) * (uint32_t)!bIsLeaf;
    orc_bvh_node* nd = &nodes[internal];
    nd->box.bounds[0].x = FLT_MAX; nd->box.bounds[0].y = FLT_MAX; nd->box.bounds[0].z = FLT_MAX;
    nd->box.bounds[1].x = FLT_MIN; nd->box.bounds[1].y = FLT_MIN; nd->box.bounds[1].z = FLT_MIN;
    nd->box.bounds[0].w = 0.0f;                       /* type = INNER_NODE */
    if (internal == numLeaves) nd->data[0] = UINT_MAX; /* root parent */
    nd->data[1] = child_A;
    nd->data[2] = child_B;
    uint32_t total = 2 * numLeaves - 1;
    /* LITERAL_ZERO: children indices can be garbage-free here, but keep the documented
     * "out-of-range parent indices skipped" rule for symmetry with refit. */
    if (!literal || child_A < total) nodes[child_A].data[0] = internal;
    if (!literal || child_B < total) nodes[child_B].data[0] = internal;
}

/* kernel buildRadixTree, BVHKernels.cl:55-60.  FIXED: guard idx < n-1 (all n-1 internal
 * nodes); LITERAL_ZERO: the guard as written, idx < n-2 (internal node n-2 is never built). */
void orc_radix_tree(orc_bvh_node* nodes, const orc_pair* sorted, uint32_t n, int mode)
{
    uint32_t limit = (mode == ORC_MODE_LITERAL_ZERO) ? n - 2 : n - 1;
    for (uint32_t idx = 0; idx < limit; idx++)
        construct_node(nodes, sorted, n, (int)idx, mode == ORC_MODE_LITERAL_ZERO);
}

/* merge3, AABB.h:107-113 over all four components (MIN4/MAX4, CLPortability.h:294-318) */
static orc_f4 min4(orc_f4 a, orc_f4 b) { orc_f4 r = { RMIN(a.x, b.x), RMIN(a.y, b.y), RMIN(a.z, b.z), RMIN(a.w, b.w) }; return r; }
static orc_f4 max4(orc_f4 a, orc_f4 b) { orc_f4 r = { RMAX(a.x, b.x), RMAX(a.y, b.y), RMAX(a.z, b.z), RMAX(a.w, b.w) }; return r; }
/* mergeBoundingBox, BVH.h:328-334 */
static void merge_bounding_box(orc_bvh_node* nodes, uint32_t cur)
{
    uint32_t a = nodes[cur].data[1], b = nodes[cur].data[2];
    orc_aabb r;
    r.bounds[0] = min4(min4(nodes[a].box.bounds[0], nodes[b].box.bounds[0]), nodes[cur].box.bounds[0]);
    r.bounds[1] = max4(max4(nodes[a].box.bounds[1], nodes[b].box.bounds[1]), nodes[cur].box.bounds[1]);
    nodes[cur].box = r;
}

/* kernel computeBoundingBoxes, BVHKernels.cl:65-79, serialised leaf by leaf.
 * FIXED: one walker per leaf, the SECOND arriver at a node merges (atomic's return value).
 * LITERAL_ZERO: walkers for gid in [0, ceil(n/64)*64) clamped to leaf n-1 as launched by
 * BVHManager.cpp:214-217 (so the last leaf's ancestors see extra arrivals), the counter is
 * re-read after the increment, out-of-range parents end the walk. */
void orc_refit(orc_bvh_node* nodes, uint32_t n, int mode)
{
    uint32_t* counters = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    uint32_t total = 2 * n - 1;
    uint32_t walkers = n;
    if (mode == ORC_MODE_LITERAL_ZERO) walkers = ((n + 63) / 64) * 64;
    for (uint32_t gid = 0; gid < walkers; gid++) {
        uint32_t current = RMIN(gid, n - 1);
        current = nodes[current].data[0];
        while (current != UINT_MAX) {
            if (current < n || current >= total) break;   /* only reachable in LITERAL_ZERO */
            uint32_t c = ++counters[current - n];
            if (c > 1) {
                merge_bounding_box(nodes, current);
                current = nodes[current].data[0];
            } else break;
        }
    }
    free(counters);
}

int orc_build_bvh(const void* scene, int mode, orc_bvh_node* nodes, orc_pair* pairs)
{
    uint32_t n = (uint32_t)sh_triangles((const char*)scene);
    if (n < 2) return 1;
    uint32_t N = orc_morton_buffer_items(n);
    memset(nodes, 0, (size_t)(2 * n - 1) * sizeof *nodes);        /* zero-initialised device buffers */
    for (uint32_t i = 0; i < N; i++) { pairs[i].key = UINT_MAX; pairs[i].value = UINT_MAX; }
    orc_morton_leaves(scene, nodes, pairs);
    orc_bitonic_sort(pairs, N);
    orc_radix_tree(nodes, pairs, n, mode);
    orc_refit(nodes, n, mode);
    return 0;
}

/* every leaf's parent chain reaches the root (index n) without cycles: Testing/BVHTest.h:52-83 */
int orc_check_hierarchy(const orc_bvh_node* nodes, uint32_t n, uint32_t* depth_out)
{
    uint32_t maxd = 0;
    for (uint32_t leaf = 0; leaf < n; leaf++) {
        uint32_t cur = leaf, d = 0;
        while (cur != n) {
            cur = nodes[cur].data[0];
            if (cur < n || cur >= 2 * n - 1) return 1;
            if (++d > 2 * n) return 2;
        }
        if (d > maxd) maxd = d;
    }
    if (nodes[n].data[0] != UINT_MAX) return 3;
    if (depth_out) *depth_out = maxd;
    return 0;
}

/* ------------------------------------------------------------------ */
/* traversal: bvh_generate_contact, BVH.h:344-408                      */
/* ------------------------------------------------------------------ */
#define ORC_STACK 4096   /* the reference array holds 32 entries and is unchecked */

void orc_traverse(const orc_ray* ray, const orc_bvh_node* bvh, uint32_t rootIdx, const void* scene_,
                  orc_contact* out, orc_stats* st)
{
    const char* scene = (const char*)scene_;
    uint32_t stack[ORC_STACK];
    uint32_t sp = 0;
    uint32_t cur = rootIdx;
    stack[sp++] = UINT_MAX;
    orc_f4 res = { 0.0f, 0.0f, 0.0f, FLT_MAX };   /* xyz uninitialised in the reference */
    uint32_t resMaterial = 0;
    uint64_t inner = 0, leaves = 0;
    uint32_t maxsp = sp;
    do {
        const orc_bvh_node* node = &bvh[cur];
        if (node->box.bounds[0].w == 0.0f) {         /* type(node) == INNER_NODE */
            uint32_t a = node->data[1], b = node->data[2];
            const orc_aabb* boxA = &bvh[a].box;
            const orc_aabb* boxB = &bvh[b].box;
            float tA = orc_aabb_intersect(boxA, ray->origin, ray->direction);
            float tB = orc_aabb_intersect(boxB, ray->origin, ray->direction);
            int aValid = tA > 0 || orc_point_inside(boxA, ray->origin);
            int bValid = tB > 0 || orc_point_inside(boxB, ray->origin);
            inner++;
            if (aValid && bValid) {
                cur = a;
                if (sp < ORC_STACK) stack[sp] = b;
                sp++;
                if (sp > maxsp) maxsp = sp;
            } else if (aValid) cur = a;
            else if (bValid) cur = b;
            else cur = stack[--sp];
        } else {
            const char* mesh = mesh_at(node->data[2], model_at(node->data[3], scene));
            uint32_t base = node->data[1] * 3;
            orc_f4 c = orc_triangle_intersect(vertex_at(index_at(base, mesh), mesh),
                                              vertex_at(index_at(base + 1, mesh), mesh),
                                              vertex_at(index_at(base + 2, mesh), mesh),
                                              ray->origin, ray->direction);
            leaves++;
            if (c.w > 0 && c.w < res.w) {
                res = c;
                resMaterial = (uint32_t)mesh_material(mesh);
            }
            cur = stack[--sp];
        }
    } while (cur != UINT_MAX);
    memset(out, 0, sizeof *out);
    out->normal_t = res;
    if (out->normal_t.w == FLT_MAX) out->normal_t.w = 0;
    out->materialIndex = resMaterial;
    if (st) {
        st->inner_visits += inner;
        st->leaf_tests += leaves;
        if (maxsp > st->max_stack) st->max_stack = maxsp;
        if (out->normal_t.w > 0) { st->hits++; st->sum_t += (double)out->normal_t.w; }
    }
}

/* ---- threaded drivers ------------------------------------------------ */
typedef struct {
    int kind;                       /* 0 primary, 1 rays, 2 brute, 3 shade */
    const orc_camera* cam;
    const orc_ray* rays;
    const orc_bvh_node* nodes;
    uint32_t n;
    const char* scene;
    orc_contact* contacts;
    const orc_contact* in_contacts;
    orc_f4* rgba;
    uint32_t first, count;
    volatile uint32_t* next;
    orc_stats st;
    uint32_t* per_ray_inner;        /* optional: inner visits of every ray (load-balance analysis) */
    /* brute-force triangle table */
    const orc_f4* tri; const uint32_t* mat; uint32_t ntri;
    char pad_[256];                 /* one job per thread: keeps the per-visit counters of two threads on different cache lines */
} job_t;

enum { CHUNK = 256 };

static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    for (;;) {
        uint32_t start = __sync_fetch_and_add(j->next, CHUNK);
        if (start >= j->count) break;
        uint32_t end = start + CHUNK; if (end > j->count) end = j->count;
        for (uint32_t k = start; k < end; k++) {
            if (j->kind == 0) {
                /* kernel generateContacts, BVHKernels.cl:84-98 */
                uint32_t gid = j->first + k;
                orc_ray r; orc_generate_ray(j->cam, gid, &r);
                uint64_t before = j->st.inner_visits;
                orc_contact c; orc_traverse(&r, j->nodes, j->n, j->scene, &c, &j->st);
                c.pixelIndex = gid;
                j->contacts[k] = c;
                if (j->per_ray_inner) j->per_ray_inner[k] = (uint32_t)(j->st.inner_visits - before);
            } else if (j->kind == 1) {
                /* kernel generateContacts2, BVHKernels.cl:103-117 */
                orc_contact c; orc_traverse(&j->rays[k], j->nodes, j->n, j->scene, &c, &j->st);
                c.pixelIndex = k;
                j->contacts[k] = c;
            } else if (j->kind == 2) {
                uint32_t gid = j->first + k;
                orc_ray r; orc_generate_ray(j->cam, gid, &r);
                orc_f4 res = { 0, 0, 0, FLT_MAX }; uint32_t m = 0;
                for (uint32_t t = 0; t < j->ntri; t++) {
                    orc_f4 c = orc_triangle_intersect(j->tri[3 * t], j->tri[3 * t + 1], j->tri[3 * t + 2], r.origin, r.direction);
                    if (c.w > 0 && c.w < res.w) { res = c; m = j->mat[t]; }
                }
                orc_contact c; memset(&c, 0, sizeof c);
                c.normal_t = res; if (c.normal_t.w == FLT_MAX) c.normal_t.w = 0;
                c.materialIndex = m; c.pixelIndex = gid;
                j->contacts[k] = c;
            } else {
                /* RTFrame shading loop, OpenCLRayTracerDemo.cpp:301-316 */
                orc_contact c = j->in_contacts[k];
                if (c.normal_t.w > 0) {
                    orc_f4 color;
                    orc_shade_blinn_phong(j->cam, &c, j->scene, &color);
                    j->rgba[c.pixelIndex] = color;
                }
            }
        }
    }
    return NULL;
}

static void run_jobs(job_t* proto, int nthreads, orc_stats* st)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile uint32_t next = 0;
    job_t* jobs = (job_t*)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) { jobs[i] = *proto; jobs[i].next = &next; memset(&jobs[i].st, 0, sizeof(orc_stats)); }
    if (nthreads == 1) worker(&jobs[0]);
    else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    if (st) {
        for (int i = 0; i < nthreads; i++) {
            st->inner_visits += jobs[i].st.inner_visits;
            st->leaf_tests += jobs[i].st.leaf_tests;
            st->hits += jobs[i].st.hits;
            st->sum_t += jobs[i].st.sum_t;
            if (jobs[i].st.max_stack > st->max_stack) st->max_stack = jobs[i].st.max_stack;
        }
    }
    free(jobs); free(th);
}

void orc_trace_primary_range(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                             uint32_t first, uint32_t count, orc_contact* contacts, orc_stats* st, int nthreads)
{
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 0; j.cam = cam; j.nodes = nodes; j.n = n; j.scene = (const char*)scene;
    j.contacts = contacts; j.first = first; j.count = count;
    run_jobs(&j, nthreads, st);
}

void orc_trace_primary_visits(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                              orc_contact* contacts, uint32_t* inner_visits, orc_stats* st, int nthreads)
{
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 0; j.cam = cam; j.nodes = nodes; j.n = n; j.scene = (const char*)scene;
    j.contacts = contacts; j.first = 0; j.count = cam->resX * cam->resY; j.per_ray_inner = inner_visits;
    run_jobs(&j, nthreads, st);
}

void orc_trace_primary(const orc_camera* cam, const orc_bvh_node* nodes, uint32_t n, const void* scene,
                       orc_contact* contacts, orc_stats* st, int nthreads)
{
    orc_trace_primary_range(cam, nodes, n, scene, 0, cam->resX * cam->resY, contacts, st, nthreads);
}

void orc_trace_rays(const orc_ray* rays, uint32_t count, const orc_bvh_node* nodes, uint32_t n,
                    const void* scene, orc_contact* contacts, orc_stats* st, int nthreads)
{
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 1; j.rays = rays; j.nodes = nodes; j.n = n; j.scene = (const char*)scene;
    j.contacts = contacts; j.count = count;
    run_jobs(&j, nthreads, st);
}

void orc_brute_primary(const orc_camera* cam, const void* scene, orc_contact* contacts, int nthreads)
{
    uint32_t n = (uint32_t)sh_triangles((const char*)scene);
    orc_f4* tri = (orc_f4*)malloc((size_t)n * 3 * sizeof(orc_f4));
    uint32_t* mat = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    for (uint32_t t = 0; t < n; t++) { uint32_t ref[3]; orc_scene_triangle(scene, t, &tri[3 * t], ref, &mat[t]); }
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 2; j.cam = cam; j.scene = (const char*)scene; j.contacts = contacts;
    j.count = cam->resX * cam->resY; j.tri = tri; j.mat = mat; j.ntri = n;
    run_jobs(&j, nthreads, NULL);
    free(tri); free(mat);
}

/* ------------------------------------------------------------------ */
/* shading: cpuShadeBlinnPhong, Shading.h:74-166                       */
/* ------------------------------------------------------------------ */

/* lightEnergyPercentage, Primitives/Light.h:55-58 */
static float light_energy_percentage(float distance, float lightEnergy)
{
    return RMAX((1.0f - distance / lightEnergy), 0.0f);
}

void orc_shade_blinn_phong(const orc_camera* cam, const orc_contact* contact, const void* scene_, orc_f4* out)
{
    const char* scene = (const char*)scene_;
    /* default material, Shading.h:76-96; Material layout Primitives/Material.h:46-58 */
    orc_f4 ambient = { 0.2f, 0.2f, 0.2f, 0 }, diffuse = { 1.0f, 1.0f, 1.0f, 0 }, specular = { 1.0f, 0.0f, 0.0f, 0 };
    float shininess = 0.4f;
    uint64_t numOfLights = sh_lights(scene);
    orc_ray r; orc_generate_ray(cam, contact->pixelIndex, &r);
    orc_f4 P = v_add(r.origin, v_scale(r.direction, contact->normal_t.w));
    if (contact->materialIndex < sh_materials(scene)) {
        const char* m = materials_ptr(scene) + (size_t)contact->materialIndex * MATERIAL_SIZE;
        ambient = rd_f4(m); diffuse = rd_f4(m + 16); specular = rd_f4(m + 32);
        memcpy(&shininess, m + 80, 4);
    }
    orc_f4 camPos = { cam->m[3], cam->m[7], cam->m[11], 0 };
    orc_f4 N = contact->normal_t;
    orc_f4 V = v_normalize(v_sub(camPos, P));
    orc_f4 color = { ambient.x, ambient.y, ambient.z, 0.0f };
    float totalEnergy = 0.0f;
    /* lights = scene lights in order, then the camera light (energy 100000) : Shading.h:112-133 */
    size_t nl = (size_t)numOfLights + 1;
    orc_f4* lights = (orc_f4*)malloc(nl * sizeof(orc_f4));
    for (uint64_t i = 0; i < numOfLights; i++) {
        orc_f4 l = rd_f4(lights_ptr(scene) + i * LIGHT_SIZE);
        orc_f4 L = v_sub(l, P);
        float dist = v_length(L);
        totalEnergy += light_energy_percentage(dist, l.w) * l.w;
        lights[i] = l;
    }
    {
        orc_f4 l = { camPos.x, camPos.y, camPos.z, 100000 };
        orc_f4 L = v_sub(l, P);
        float dist = v_length(L);
        totalEnergy += light_energy_percentage(dist, l.w) * l.w;
        lights[numOfLights] = l;
    }
    for (size_t i = 0; i < nl; i++) {
        orc_f4 l = lights[i];
        orc_f4 L = v_sub(l, P);
        float dist = v_length(L);
        float lightIntensity = light_energy_percentage(dist, l.w);
        float percentageOfTotal = (lightIntensity * l.w) / totalEnergy;
        float lsEnergy = lightIntensity * percentageOfTotal;
        L = v_normalize(L);
        float LDotN = v_dot(L, N);
        LDotN = LDotN > 0.0f ? LDotN : 0.0f;
        color.x += diffuse.x * LDotN * lsEnergy;
        color.y += diffuse.y * LDotN * lsEnergy;
        color.z += diffuse.z * LDotN * lsEnergy;
        orc_f4 R = v_normalize(v_sub(v_scale(N, 2.0f * LDotN), L));
        float nSpecular = powf(v_dot(R, V), shininess);
        float px = specular.x * nSpecular * lsEnergy;
        float py = specular.y * nSpecular * lsEnergy;
        float pz = specular.z * nSpecular * lsEnergy;
        color.x += RMAX(px, 0);
        color.y += RMAX(py, 0);
        color.z += RMAX(pz, 0);
    }
    free(lights);
    color.x = RMIN(color.x, 1.0f);
    color.y = RMIN(color.y, 1.0f);
    color.z = RMIN(color.z, 1.0f);
    color.w = 1.0f;
    *out = color;
}

void orc_shade_frame(const orc_camera* cam, const orc_contact* contacts, uint32_t count,
                     const void* scene, orc_f4* rgba, int nthreads)
{
    memset(rgba, 0, (size_t)cam->resX * cam->resY * sizeof(orc_f4));   /* OpenCLRayTracerDemo.cpp:298 */
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 3; j.cam = cam; j.scene = (const char*)scene; j.in_contacts = contacts; j.rgba = rgba; j.count = count;
    run_jobs(&j, nthreads, NULL);
}

/* inclusive scan: out[i] = sum_{j<=i} in[j] (PrefixSumKernels.cl:28-106; Testing/PrefixSumTest.h:51-60) */
void orc_scan_inclusive_u32(const uint32_t* in, uint32_t* out, uint32_t n)
{
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n; i++) { acc += in[i]; out[i] = acc; }
}

/* ------------------------------------------------------------------ */
/* Secondary-ray generation (EXTENSION: the reference only plans it, KNOWN_ISSUES_AND_TODO.txt 2.2: "generating rays from   */
/* viewing rays-scene contacts, and execute ray tracing kernel in loop").  This is the specification the product's          */
/* rt_generate_secondary_rays follows, operation for operation; there is no reference code to restate.                      */
/*   hit point      P = o + d * t                      (as Shading.h:93 computes it)                                         */
/*   facing normal  Nf = dot(N, d) > 0 ? -N : N                                                                             */
/*   origin         P + Nf * eps                                                                                            */
/*   SHADOW:  direction = normalize(target - origin')   with origin' = the offset origin                                    */
/*   REFLECT: direction = normalize(d - N * (2 * dot(d, N)))                                                                */
/*   idx = contact.pixelIndex; contacts without a hit (t <= 0) emit nothing; output order = contact order                   */
/* ------------------------------------------------------------------ */
uint32_t orc_generate_secondary_rays(const orc_camera* cam, const orc_ray* rays_in, const orc_contact* contacts, uint32_t count,
                                     int kind, orc_f4 target, float eps, orc_ray* out)
{
    uint32_t n = 0;
    for (uint32_t k = 0; k < count; k++) {
        const orc_contact* c = &contacts[k];
        float t = c->normal_t.w;
        if (!(t > 0.0f)) continue;
        orc_ray in;
        if (cam) orc_generate_ray(cam, c->pixelIndex, &in); else in = rays_in[k];
        orc_f4 N = { c->normal_t.x, c->normal_t.y, c->normal_t.z, 0.0f };
        orc_f4 P = v_add(in.origin, v_scale(in.direction, t));
        float dn = v_dot(N, in.direction);
        orc_f4 Nf = dn > 0.0f ? v_scale(N, -1.0f) : N;
        orc_f4 O = v_add(P, v_scale(Nf, eps));
        orc_f4 D;
        if (kind == 0) D = v_normalize(v_sub(target, O));
        else D = v_normalize(v_sub(in.direction, v_scale(N, 2.0f * v_dot(in.direction, N))));
        memset(&out[n], 0, sizeof out[n]);
        out[n].idx = c->pixelIndex;
        out[n].origin = O; out[n].direction = D;
        out[n].origin.w = 0.0f; out[n].direction.w = 0.0f;
        n++;
    }
    return n;
}
