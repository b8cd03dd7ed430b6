// Micro-benchmark: 64 lanes of a wave each need a different 64-byte record of an L2-resident table (the access pattern of
// the lane-per-ray and wave-per-ray traversal kernels).
//   A: every lane loads its record with four 16-byte loads (64 different cache lines per load instruction)
//   B: four lanes share one record per pass (each lane one 16-byte quarter, 16 records = 16 lines per load instruction),
//      four passes, quarters staged through LDS so that every lane ends up with its whole record
// Dependent chain like a traversal: the next record index comes from the record just read.
//   hipcc --offload-arch=gfx950 -O3 -o record_fetch record_fetch.hip && ./record_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((aligned(16))) Rec { float4 q[4]; };

template <int MODE>
__global__ __launch_bounds__(256) void k_fetch(const Rec* __restrict__ tab, uint32_t n_rec, uint32_t steps, float* out)
{
    __shared__ float4 stage[4][64 * 4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4* st = stage[wave];
    uint32_t node = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    float acc = 0.f;
    for (uint32_t s = 0; s < steps; s++) {
        float4 q0, q1, q2, q3;
        if (MODE == 0) {
            const float4* p = tab[node].q;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
        } else {
            float4 x[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t v = (uint32_t)p * 16u + (lane >> 2);
                const uint32_t nd = (uint32_t)__shfl((int)node, (int)v, 64);
                x[p] = tab[nd].q[lane & 3u];
            }
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t v = (uint32_t)p * 16u + (lane >> 2);
                st[v * 4u + (((lane & 3u) + (v >> 2)) & 3u)] = x[p];
            }
            __builtin_amdgcn_wave_barrier();
            q0 = st[lane * 4u + ((0u + (lane >> 2)) & 3u)];
            q1 = st[lane * 4u + ((1u + (lane >> 2)) & 3u)];
            q2 = st[lane * 4u + ((2u + (lane >> 2)) & 3u)];
            q3 = st[lane * 4u + ((3u + (lane >> 2)) & 3u)];
            __builtin_amdgcn_wave_barrier();
        }
        // a little arithmetic like a slab test, then the next node from the record
        float t = (q0.x - 1.f) * 0.5f + (q1.y - 2.f) * 0.25f + (q2.z - 3.f) * 0.125f + q3.x;
        acc += t;
        node = (__float_as_uint(q0.w) + (t > 1e30f ? 1u : 0u)) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t n_rec = argc > 1 ? atoi(argv[1]) : 100000;       // 6.4 MB: L2 / MALL resident like the BVH records
    const uint32_t steps = 200;
    std::vector<Rec> h(n_rec);
    uint32_t seed = 12345;
    for (uint32_t i = 0; i < n_rec; i++) {
        for (int j = 0; j < 4; j++) h[i].q[j] = make_float4((float)i, (float)j, 1.f, 0.f);
        seed = seed * 1664525u + 1013904223u;
        uint32_t next = seed % n_rec;
        h[i].q[0].w = __builtin_bit_cast(float, next);
    }
    Rec* d; float* out;
    const int blocks = 256 * 5;
    CHECK(hipMalloc(&d, sizeof(Rec) * n_rec)); CHECK(hipMalloc(&out, blocks * 256 * 4));
    CHECK(hipMemcpy(d, h.data(), sizeof(Rec) * n_rec, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ref(blocks * 256), got(blocks * 256);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_fetch<0>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            else hipLaunchKernelGGL(k_fetch<1>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double wave_steps = (double)blocks * 4 * steps;
            printf("mode %c rep %d: %.3f ms, %.1f ns per wave step per CU (256 CUs), %.0f GB/s of records\n", mode ? 'B' : 'A', rep, ms,
                   ms * 1e6 / (wave_steps / 256.0), wave_steps * 4096.0 / (ms * 1e-3) / 1e9);
        }
        CHECK(hipMemcpy(mode ? got.data() : ref.data(), out, blocks * 256 * 4, hipMemcpyDeviceToHost));
    }
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); i++) bad += ref[i] != got[i];
    printf("results differ in %zu of %zu lanes\n", bad, ref.size());
    return bad != 0;
}
