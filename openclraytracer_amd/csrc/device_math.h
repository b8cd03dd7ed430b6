// device_math.h -- the reference's host-branch arithmetic, restated for gfx950 device code.
//
// Everything here must produce the same binary32 results as the x86-64 oracle:
//  * translation units are compiled with -ffp-contract=off (no FMA formation);
//  * `/` and sqrtf are the correctly rounded forms (hipcc default
//    -fhip-fp32-correctly-rounded-divide-sqrt; never -ffast-math);
//  * min/max are the TERNARY macros of CLData/CLPortability.h:147-148, not v_min/v_max:
//    they differ for NaN and for signed zeros.
#pragma once
// Dual target: the device code of librtmi.so (hipcc) and, for cpuShadeBlinnPhong of the host mirror (csrc/host, g++
// -ffp-contract=off), the same functions as plain inline C++ -- one statement of the arithmetic for both.
#if defined(__HIP__) || defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_FN static __device__ __forceinline__      // (static: trace_kernels.hip is compiled twice with different bodies of dot / cross)
#else
#include <math.h>
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 v = { x, y, z, w }; return v; }
#define RT_FN static inline
#endif
#include <float.h>
#include <stdint.h>

#define RMIN(X, Y) (((X) < (Y)) ? (X) : (Y))
#define RMAX(X, Y) (((X) > (Y)) ? (X) : (Y))

struct f3 { float x, y, z; };

RT_FN f3 f3_sub(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_FN f3 f3_add(f3 a, f3 b) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_FN f3 f3_scale(f3 a, float s) { return f3{ a.x * s, a.y * s, a.z * s }; }
// cross / dot, CLPortability.h:266-278 (the HOST branch: separate multiplies and adds -- what the library computes).
// -DRTMI_DEVICE_ARITH=1 (the second compilation of trace_kernels.hip, RT_MODE_DEVICE_ARITH): the DEVICE branch leaves dot() / cross() to the OpenCL C library
// (CLPortability.h:41-45), and ROCm 7.2's (amdgcn/bitcode/opencl.bc, _Z3dotDv3_fS_ / _Z5crossDv3_fS_) fuses them as below; the
// oracle restates the same behind orc_set_device_arith (oracle/rt_oracle.c).
#if defined(RTMI_DEVICE_ARITH) && RTMI_DEVICE_ARITH
RT_FN f3 f3_cross(f3 a, f3 b)
{
    return f3{ __builtin_fmaf(a.y, b.z, b.y * -a.z), __builtin_fmaf(a.z, b.x, b.z * -a.x), __builtin_fmaf(a.x, b.y, b.x * -a.y) };
}
RT_FN float f3_dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
#else
RT_FN f3 f3_cross(f3 a, f3 b)
{
    return f3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
RT_FN float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
#endif
// cpu_normalize, CLPortability.h:151-159
RT_FN f3 f3_normalize(f3 v)
{
    float l = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return f3{ v.x * l, v.y * l, v.z * l };
}
// LENGTH3, CLPortability.h:280
RT_FN float f3_length(f3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

RT_FN f3 f3_from4(float4 v) { return f3{ v.x, v.y, v.z }; }

#if defined(__HIP__) || defined(__HIPCC__)
// a frame's contacts and colours are written once and read by nobody before the frame ends: streamed (`nt`), so that the 66 MB of a
// 1080p frame do not push the records the walk lives on out of L2 (round 3: Tenryuu 0.324 -> 0.295 ms, ManFigure 0.270 -> 0.252,
// bench 21 050 -> 21 540 Mrays/s)
typedef float v4f_stream __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_stream(void* p, float4 v)
{
    v4f_stream x = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(x, reinterpret_cast<v4f_stream*>(p));
}
__device__ __forceinline__ void store4_stream(float* p, float v) { __builtin_nontemporal_store(v, p); }
// ... and what is read once (k_shade's contacts)
__device__ __forceinline__ float4 load16_stream(const void* p)
{
    const v4f_stream x = __builtin_nontemporal_load(reinterpret_cast<const v4f_stream*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
}
#endif

// generateRay, CLStructs.h:95-106 with transformVectorByMatrix_const (Transform.h:282-295):
// the translation column is added to the direction; origin = translation.
// x = pixelIndex % resX, y = pixelIndex / resX
RT_FN void generate_ray_xy(const rt_camera& cam, uint32_t x, uint32_t y, f3& origin, f3& dir)
{
    float dx = -((float)x - (float)cam.resX * 0.5f);
    float dy = (float)y - (float)cam.resY * 0.5f;
    float dz = cam.FOVDistance;
    const float* m = cam.viewTransform;
    f3 v;
    v.x = dx * m[0] + dy * m[1] + dz * m[2] + m[3];
    v.y = dx * m[4] + dy * m[5] + dz * m[6] + m[7];
    v.z = dx * m[8] + dy * m[9] + dz * m[10] + m[11];
    dir = f3_normalize(v);
    origin = f3{ m[3], m[7], m[11] };
}
RT_FN void generate_ray(const rt_camera& cam, uint32_t pixelIndex, f3& origin, f3& dir)
{
    generate_ray_xy(cam, pixelIndex % cam.resX, pixelIndex / cam.resX, origin, dir);
}
