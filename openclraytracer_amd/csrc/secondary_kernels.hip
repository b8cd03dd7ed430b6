// secondary_kernels.hip -- rays for the next bounce, generated on the device from the contacts of the previous one
// (EXTENSION: the reference only plans it, KNOWN_ISSUES_AND_TODO.txt 2.2; the specification is oracle/rt_oracle.c,
// orc_generate_secondary_rays, followed here operation for operation).  Contacts without a hit emit nothing; the output
// keeps contact order (flags -> inclusive scan -> scatter), so the result is deterministic.
#include "rtmi_internal.h"
#include "device_math.h"

__global__ __launch_bounds__(256) void k_secondary_flags(const rt_contact* __restrict__ contacts, uint32_t count, uint32_t* __restrict__ flags)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) flags[k] = contacts[k].normalAndDistance[3] > 0.0f ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_secondary_emit(rt_camera cam, int use_cam, const rt_ray* __restrict__ rays_in,
                                                        const rt_contact* __restrict__ contacts, uint32_t count,
                                                        const uint32_t* __restrict__ scan, int kind, float tx, float ty, float tz,
                                                        float eps, rt_ray* __restrict__ out)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const float4* cp = reinterpret_cast<const float4*>(contacts + k);
    const float4 head = cp[0], nt = cp[1];
    const float t = nt.w;
    if (!(t > 0.0f)) return;
    const uint32_t pixel = __float_as_uint(head.x);
    f3 o, d;
    if (use_cam) generate_ray(cam, pixel, o, d);
    else {
        const float4* rp = reinterpret_cast<const float4*>(rays_in + k);
        const float4 ro = rp[1], rd = rp[2];
        o = f3{ ro.x, ro.y, ro.z }; d = f3{ rd.x, rd.y, rd.z };
    }
    const f3 N = f3{ nt.x, nt.y, nt.z };
    const f3 P = f3_add(o, f3_scale(d, t));
    const float dn = f3_dot(N, d);
    const f3 Nf = dn > 0.0f ? f3_scale(N, -1.0f) : N;
    const f3 O = f3_add(P, f3_scale(Nf, eps));
    f3 D;
    if (kind == RT_RAYS_SHADOW) D = f3_normalize(f3_sub(f3{ tx, ty, tz }, O));
    else D = f3_normalize(f3_sub(d, f3_scale(N, 2.0f * f3_dot(d, N))));
    float4* op = reinterpret_cast<float4*>(out + (scan[k] - 1u));
    const uint4 h = make_uint4(pixel, 0u, 0u, 0u);
    op[0] = *reinterpret_cast<const float4*>(&h);
    op[1] = make_float4(O.x, O.y, O.z, 0.0f);
    op[2] = make_float4(D.x, D.y, D.z, 0.0f);
}

// the camera's own rays as a ray buffer (generateRay, CLStructs.h:95-106, one per pixel; Ray.idx = pixel): what a host hands to
// generateContacts(CLBuffer& rays, ..) (BVHManager.cpp:292-329) when it wants the first bounce through the ray-buffer entry
__global__ __launch_bounds__(256) void k_primary_rays(rt_camera cam, uint32_t count, rt_ray* __restrict__ out)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    f3 o, d;
    generate_ray(cam, k, o, d);
    float4* op = reinterpret_cast<float4*>(out + k);
    const uint4 h = make_uint4(k, 0u, 0u, 0u);
    op[0] = *reinterpret_cast<const float4*>(&h);
    op[1] = make_float4(o.x, o.y, o.z, 0.0f);
    op[2] = make_float4(d.x, d.y, d.z, 0.0f);
}

void launch_primary_rays(hipStream_t s, const rt_camera* cam, rt_ray* out)
{
    const uint32_t count = cam->resX * cam->resY;
    hipLaunchKernelGGL(k_primary_rays, dim3((count + 255) / 256), dim3(256), 0, s, *cam, count, out);
}

void launch_secondary_flags(hipStream_t s, const rt_contact* contacts, uint32_t count, uint32_t* flags)
{
    hipLaunchKernelGGL(k_secondary_flags, dim3((count + 255) / 256), dim3(256), 0, s, contacts, count, flags);
}

void launch_secondary_emit(hipStream_t s, const rt_camera* cam, const rt_ray* rays_in, const rt_contact* contacts, uint32_t count,
                           const uint32_t* scan, int kind, const float target[3], float eps, rt_ray* out)
{
    rt_camera c{};
    if (cam) c = *cam;
    hipLaunchKernelGGL(k_secondary_emit, dim3((count + 255) / 256), dim3(256), 0, s, c, cam ? 1 : 0, rays_in, contacts, count, scan,
                       kind, target[0], target[1], target[2], eps, out);
}

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_secondary()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_secondary_flags)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
