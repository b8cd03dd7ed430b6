// CLData/CLStructs.h -- the PODs and host helpers of the reference that cross the manager boundary
// (OpenCLRayTracer/Include/CLData/CLStructs.h:44-118, Transform.h:44-154,397-413), byte-identical to
// include/rtmi.h's rt_camera / rt_ray / rt_contact.  Global namespace, as in the reference.
#ifndef RTMI_HOST_CLSTRUCTS_H
#define RTMI_HOST_CLSTRUCTS_H
#include <cmath>
#include <cstdint>

struct alignas(16) CL_FLOAT4_ { float x, y, z, w; };
typedef CL_FLOAT4_ CL_FLOAT4;
typedef CL_FLOAT4_ CL_FLOAT3;      // cl_float3 is a 16-byte type on the host
typedef float CL_FLOAT;
typedef uint32_t CL_UINT;
struct alignas(16) CL_UINT3 { uint32_t x, y, z, w; };   // cl_uint3 is a 16-byte type on the host

struct alignas(16) AABB { CL_FLOAT4 bounds[2]; };                       // Primitives/AABB.h:47-52
struct alignas(16) Matrix4 { float data[12]; };                       // Transform.h:150-154
struct alignas(16) Quaternion { float r, i, j, k; };                  // Transform.h:44-70

struct alignas(16) Camera {                                           // CLStructs.h:44-51
    float FOVDistance;
    uint32_t resX, resY, supersampilngFactor;
    Matrix4 viewTransform;
};
struct alignas(16) Ray { uint32_t idx; CL_FLOAT3 origin; CL_FLOAT3 direction; };          // CLStructs.h:82-87
struct alignas(16) Contact {                                          // CLStructs.h:112-118
    uint32_t pixelIndex, materialIndex, pad[2];
    CL_FLOAT4 normalAndintersectionDistance;
};
#define contactDist normalAndintersectionDistance.w
static_assert(sizeof(Camera) == 64 && sizeof(Ray) == 48 && sizeof(Contact) == 32, "POD layout");

const float PI = 3.14159265359f;                                      // CLStructs.h:60-61
const float DEG2RAD = PI / 180.0f;
inline float FOVDistFromAngle(float angle, float /*resx*/, float resy)   // CLStructs.h:70-74
{
    float a = resy * 0.5f;
    return a / std::tan(angle * 0.5f * DEG2RAD);
}
inline Matrix4 identityTransform()                                    // Transform.h:160-166
{
    Matrix4 m;
    for (int i = 0; i < 12; i++) m.data[i] = 0;
    m.data[0] = m.data[5] = m.data[10] = 1;
    return m;
}
inline Quaternion zeroRotation() { Quaternion q; q.r = 1.0f; q.i = q.j = q.k = 0.0f; return q; }   // Transform.h:75-81
inline void normalizeQuaternion(Quaternion* q)                        // Transform.h:88-106
{
    float d = q->r * q->r + q->i * q->i + q->j * q->j + q->k * q->k;
    if (d < 1.1920928955078125e-07f) { q->i = q->j = q->k = 0.0f; q->r = 1; return; }
    d = 1.0f / std::sqrt(d);
    q->r *= d; q->i *= d; q->j *= d; q->k *= d;
}
inline void setOrientationAndPos(Matrix4* t, const Quaternion& q, CL_FLOAT3 pos)   // Transform.h:397-413
{
    t->data[0] = 1 - (2 * q.j * q.j + 2 * q.k * q.k); t->data[1] = 2 * q.i * q.j + 2 * q.k * q.r;
    t->data[2] = 2 * q.i * q.k - 2 * q.j * q.r;       t->data[3] = pos.x;
    t->data[4] = 2 * q.i * q.j - 2 * q.k * q.r;       t->data[5] = 1 - (2 * q.i * q.i + 2 * q.k * q.k);
    t->data[6] = 2 * q.j * q.k + 2 * q.i * q.r;       t->data[7] = pos.y;
    t->data[8] = 2 * q.i * q.k + 2 * q.j * q.r;       t->data[9] = 2 * q.j * q.k - 2 * q.i * q.r;
    t->data[10] = 1 - (2 * q.i * q.i + 2 * q.j * q.j); t->data[11] = pos.z;
}
#endif
