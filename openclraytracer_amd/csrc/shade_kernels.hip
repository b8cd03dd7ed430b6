// shade_kernels.hip -- primary-ray Blinn-Phong shading on the device (gfx950).
//
// The reference shades on the HOST: RTFrame copies every Contact back and loops over them calling
// cpuShadeBlinnPhong (src/Demos/OpenCLRayTracerDemo/OpenCLRayTracerDemo.cpp:298-316,
// Include/CLData/Shading.h:74-166, lightEnergyPercentage Primitives/Light.h:55-58).  This kernel
// restates that function per contact; pixels without a hit get (0,0,0,0) (the memset at :298).
// powf comes from the device library, hence the 1e-5 relative tolerance on RGB.
#include "rtmi_internal.h"
#include "device_math.h"
#include "shade_math.h"

struct ShadeArgs {
    ShadeScene sc;
    rt_camera cam;
    const rt_contact* __restrict__ contacts;
    uint32_t count;
    float4* __restrict__ rgba;
    uint32_t tiled, rank, nranks, tiles_x;
    uint32_t plane_stride;                     // != 0: three float planes (r with the hit flag in its sign bit, g, b) of this many slots
};

__global__ __launch_bounds__(256) void k_shade(ShadeArgs a)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.count) return;
    // contacts are 32 B: two float4 loads
    const float4* p = reinterpret_cast<const float4*>(a.contacts + i);
    const float4 h = load16_stream(p), nt = load16_stream(p + 1);
    rt_contact c;
    c.pixelIndex = __float_as_uint(h.x); c.materialIndex = __float_as_uint(h.y); c.pad[0] = 0; c.pad[1] = 0;
    c.normalAndDistance[0] = nt.x; c.normalAndDistance[1] = nt.y; c.normalAndDistance[2] = nt.z; c.normalAndDistance[3] = nt.w;
    // untiled: exactly RTFrame -- the frame was cleared by the launcher (memset at :298) and only hits are
    //          stored, at colors[contact.pixelIndex] (:305-311);
    // tiled  : compact slot i keeps its slot, misses and out-of-frame slots store zeros.
    if (a.plane_stride) {
        // packed compact buffer for the gather: 12 bytes per slot instead of 16.  Alpha is 1 for a hit and 0 for a miss
        // (Shading.h:164 / the cleared frame) and colours are clamped to [0, 1], so the hit flag travels in the sign bit of
        // the red plane (a hit with red 0 is -0.0f) and rt_untile_rgba_packed restores RGBA bit for bit.
        float* planes = reinterpret_cast<float*>(a.rgba);
        float4 col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (nt.w > 0.0f) { col = shade_one(a.sc, a.cam, c); col.x = __uint_as_float(__float_as_uint(col.x) | 0x80000000u); }
        store4_stream(planes + i, col.x);
        store4_stream(planes + (size_t)a.plane_stride + i, col.y);
        store4_stream(planes + 2u * (size_t)a.plane_stride + i, col.z);
        return;
    }
    if (nt.w > 0.0f) store16_stream(a.rgba + (a.tiled ? i : c.pixelIndex), shade_one(a.sc, a.cam, c));
    else if (a.tiled) store16_stream(a.rgba + i, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
}

void launch_shade(hipStream_t s, const DeviceScene& sc, const rt_camera& cam, const rt_contact* contacts,
                  uint32_t count, float4* rgba, bool tiled, uint32_t rank, uint32_t nranks, uint32_t plane_stride,
                  GraphNodes* graph_record, GraphNodes* graph_update)
{
    if (!count) return;
    ShadeArgs a{};
    a.sc.scene = sc.bytes;
    a.sc.n_lights = sc.n_lights; a.sc.n_spheres = sc.n_spheres; a.sc.n_materials = sc.n_materials;
    a.cam = cam;
    a.contacts = contacts; a.count = count; a.rgba = rgba;
    a.tiled = tiled ? 1u : 0u; a.rank = rank; a.nranks = nranks; a.tiles_x = rt_tiles_x(cam.resX); a.plane_stride = plane_stride;
    if (!tiled && !graph_update) (void)hipMemsetAsync(rgba, 0, (size_t)cam.resX * cam.resY * sizeof(float4), s);   // memset(colors,0), :298
    rt_launch_node(graph_record, graph_update, RT_NODE_SHADE, k_shade, dim3((count + 255) / 256), dim3(256), s, a);
}

// gathered[r * slots_per_rank + ordinal*4096 + ly*64 + lx]  ->  frame[y*resX + x]
__global__ __launch_bounds__(256) void k_untile_rgba(rt_camera cam, uint32_t nranks, const float4* __restrict__ gathered,
                                                     uint32_t slots_per_rank, float4* __restrict__ frame)
{
    uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= cam.resX * cam.resY) return;
    uint32_t y = pixel / cam.resX, x = pixel - y * cam.resX;
    uint32_t tx = x / RT_TILE, ty = y / RT_TILE;
    uint32_t tile = ty * rt_tiles_x(cam.resX) + tx;
    uint32_t rank, ordinal;
    rt_tile_owner(tile, nranks, rank, ordinal);
    uint32_t slot = ordinal * (RT_TILE * RT_TILE) + (y - ty * RT_TILE) * RT_TILE + (x - tx * RT_TILE);
    frame[pixel] = gathered[(uint64_t)rank * slots_per_rank + slot];
}

// the same from the packed planes of rt_shade_blinn_phong_tiled_packed: gathered[r][plane][slot]
__global__ __launch_bounds__(256) void k_untile_rgba_packed(rt_camera cam, uint32_t nranks, const float* __restrict__ gathered,
                                                            uint32_t slots_per_rank, uint64_t rank_stride, float4* __restrict__ frame)
{
    uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= cam.resX * cam.resY) return;
    uint32_t y = pixel / cam.resX, x = pixel - y * cam.resX;
    uint32_t tx = x / RT_TILE, ty = y / RT_TILE;
    uint32_t tile = ty * rt_tiles_x(cam.resX) + tx;
    uint32_t rank, ordinal;
    rt_tile_owner(tile, nranks, rank, ordinal);
    uint32_t slot = ordinal * (RT_TILE * RT_TILE) + (y - ty * RT_TILE) * RT_TILE + (x - tx * RT_TILE);
    const float* g = gathered + (uint64_t)rank * rank_stride + slot;
    const uint32_t r = __float_as_uint(g[0]);
    frame[pixel] = make_float4(__uint_as_float(r & 0x7FFFFFFFu), g[slots_per_rank], g[2u * (uint64_t)slots_per_rank],
                               (r & 0x80000000u) ? 1.0f : 0.0f);
}

void launch_untile_rgba_packed(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float* gathered,
                               uint32_t slots_per_rank, uint64_t rank_stride, float4* frame)
{
    uint32_t n = cam.resX * cam.resY;
    hipLaunchKernelGGL(k_untile_rgba_packed, dim3((n + 255) / 256), dim3(256), 0, s, cam, nranks, gathered, slots_per_rank, rank_stride, frame);
}

// the multi-GPU frame driver's variant (rtmi_multi.hip): rank 0's planes are read where its shading kernel left them (its own
// send buffer, plane stride root_slots: nothing of rank 0 is copied), the peers' from the gather buffer, peer r at
// peers + (r - 1) * peer_stride with plane stride peer_slots
__global__ __launch_bounds__(256) void k_untile_rgba_split(rt_camera cam, uint32_t nranks, const float* __restrict__ root_planes, uint32_t root_slots,
                                                           const float* __restrict__ peers, uint32_t peer_slots, uint64_t peer_stride,
                                                           float4* __restrict__ frame)
{
    uint32_t pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= cam.resX * cam.resY) return;
    uint32_t y = pixel / cam.resX, x = pixel - y * cam.resX;
    uint32_t tx = x / RT_TILE, ty = y / RT_TILE;
    uint32_t tile = ty * rt_tiles_x(cam.resX) + tx;
    uint32_t rank, ordinal;
    rt_tile_owner(tile, nranks, rank, ordinal);
    uint32_t slot = ordinal * (RT_TILE * RT_TILE) + (y - ty * RT_TILE) * RT_TILE + (x - tx * RT_TILE);
    const float* g = rank == 0u ? root_planes + slot : peers + (uint64_t)(rank - 1u) * peer_stride + slot;
    const uint64_t plane = rank == 0u ? root_slots : peer_slots;
    const uint32_t r = __float_as_uint(g[0]);
    frame[pixel] = make_float4(__uint_as_float(r & 0x7FFFFFFFu), g[plane], g[2u * plane], (r & 0x80000000u) ? 1.0f : 0.0f);
}

void launch_untile_rgba_split(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float* root_planes, uint32_t root_slots,
                              const float* peers, uint32_t peer_slots, uint64_t peer_stride, float4* frame)
{
    uint32_t n = cam.resX * cam.resY;
    hipLaunchKernelGGL(k_untile_rgba_split, dim3((n + 255) / 256), dim3(256), 0, s, cam, nranks, root_planes, root_slots, peers, peer_slots,
                       peer_stride, frame);
}

void launch_untile_rgba(hipStream_t s, const rt_camera& cam, uint32_t nranks, const float4* gathered,
                        uint32_t slots_per_rank, float4* frame)
{
    uint32_t n = cam.resX * cam.resY;
    hipLaunchKernelGGL(k_untile_rgba, dim3((n + 255) / 256), dim3(256), 0, s, cam, nranks, gathered, slots_per_rank,
                       frame);
}

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_shade()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_shade)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
