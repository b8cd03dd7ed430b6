// micro test: one wave-instruction of LDS-DMA (global_load_lds_dwordx4, lanes 0-3, saddr + 32-bit voffset) fetches a 64-byte record
// into the wave's LDS slot; all lanes read it back by broadcast.   hipcc --offload-arch=gfx950 -O3 glds_leaf.hip -o glds_leaf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct Rec { float v[16]; };

__device__ __forceinline__ void dma_record(const Rec* base, uint32_t id, uint32_t lane, void* lds_slot)
{
    const uint32_t voff = (id << 6) + ((lane & 3u) << 4);
    const uint32_t lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_slot);
    uint32_t keep_m0; uint64_t keep_exec;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b64 %1, exec\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_mov_b64 exec, 0xf\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0), "=&s"(keep_exec) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(256) void k(const Rec* recs, const uint32_t* ids, float* out, uint32_t rounds)
{
    __shared__ float4 s_tri[4][16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wave;
    float acc = 0.f;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t id0 = __builtin_amdgcn_readfirstlane(ids[(gw * rounds + r) * 2]), id1 = __builtin_amdgcn_readfirstlane(ids[(gw * rounds + r) * 2 + 1]);
        dma_record(recs, id0, lane, &s_tri[wave][0]);
        dma_record(recs, id1, lane, &s_tri[wave][4]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float4 a = s_tri[wave][lane & 3u], b = s_tri[wave][4 + (lane & 3u)];
        acc += a.x + a.y * 2.f + a.z * 3.f + a.w * 4.f + b.x * 5.f + b.y * 6.f + b.z * 7.f + b.w * 8.f;
        __builtin_amdgcn_wave_barrier();
    }
    out[gw * 64 + lane] = acc;
}

int main()
{
    const uint32_t NREC = 4096, WAVES = 1024, ROUNDS = 16;
    std::vector<Rec> h(NREC);
    for (uint32_t i = 0; i < NREC; i++) for (int k = 0; k < 16; k++) h[i].v[k] = (float)((i * 16 + k) % 977);
    std::vector<uint32_t> ids(WAVES * ROUNDS * 2);
    uint32_t x = 12345;
    for (auto& v : ids) { x = x * 1664525u + 1013904223u; v = (x >> 8) % NREC; }
    Rec* d; uint32_t* di; float* dout;
    hipMalloc(&d, NREC * sizeof(Rec)); hipMalloc(&di, ids.size() * 4); hipMalloc(&dout, WAVES * 64 * 4);
    hipMemcpy(d, h.data(), NREC * sizeof(Rec), hipMemcpyHostToDevice); hipMemcpy(di, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(WAVES / 4), dim3(256), 0, 0, d, di, dout, ROUNDS);
    std::vector<float> o(WAVES * 64);
    hipError_t e = hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
    uint32_t bad = 0;
    for (uint32_t w = 0; w < WAVES; w++) for (uint32_t l = 0; l < 64; l++) {
        float acc = 0.f;
        for (uint32_t r = 0; r < ROUNDS; r++) {
            const Rec& a = h[ids[(w * ROUNDS + r) * 2]]; const Rec& b = h[ids[(w * ROUNDS + r) * 2 + 1]];
            const int p = (l & 3) * 4;
            acc += a.v[p] + a.v[p + 1] * 2.f + a.v[p + 2] * 3.f + a.v[p + 3] * 4.f + b.v[p] * 5.f + b.v[p + 1] * 6.f + b.v[p + 2] * 7.f + b.v[p + 3] * 8.f;
        }
        if (acc != o[w * 64 + l]) bad++;
    }
    printf("glds_leaf: %u mismatches of %u\n", bad, WAVES * 64);
    return bad != 0;
}
