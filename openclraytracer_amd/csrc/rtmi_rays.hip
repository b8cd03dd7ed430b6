// rtmi_rays.hip -- C ABI, the ray-buffer entry point: generateContacts(CLBuffer& rays, CLBuffer& contacts, n), BVHManager.cpp:292-329
// (kernel generateContacts2, BVHKernels.cl:103-117).
#include "rtmi_internal.h"
#include "rtmi_host.h"
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <sstream>

extern "C" {

int rt_bvh_generate_contacts_rays(rt_bvh* b, const void* d_rays, uint32_t ray_count, void* d_contacts)
{
    if (!b) return RT_FAIL(nullptr, "bvh is null");
    rt_ctx* ctx = b->ctx;
    if (!b->built) return RT_FAIL(ctx, "construct must precede generateContacts");
    if (ray_count == 0) return RT_SUCCESS;
    if (!d_rays || !d_contacts) return RT_FAIL(ctx, "null buffer");
    RT_ENTER(ctx);
    if (ensure_long_list(ctx, ray_count)) return RT_ERROR;
    launch_trace_rays(ctx->stream, make_launch(b), (const rt_ray*)d_rays, ray_count, (rt_contact*)d_contacts);
    RT_HIP(ctx, hipGetLastError());
    return RT_SUCCESS;
}

}  // extern "C"
