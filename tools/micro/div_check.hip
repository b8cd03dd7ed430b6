// div_check.hip -- which short division / reciprocal sequences give the correctly rounded IEEE result on gfx950, and on which operands:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/micro/div_check tools/micro/div_check.hip
// (a) 1/x by v_rcp_f32 + one Newton step (two FMAs), every one of the 2^32 operands;
// (b) x / (float)res with r = RN(1 / (float)res) given (IEEE division, computed once): q0 = x r, e = fma(-y, q0, x), q1 = fma(e, r, q0)
//     ("three operations") and one more round of the last two ("five operations"), every x, res = 1 .. RES_MAX;
// (c) the same two sequences on 2^14 arbitrary divisors y in [2^-24, 2^24) x 2^22 arbitrary x.
// A result counts as equal when the bits are equal or both are NaN.  Mismatches are counted per class of the operand x:
// [0] |x| in [2^-90, 2^90], [1] smaller non-zero, [2] larger finite, [3] zero, [4] infinite, [5] NaN.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#define RES_MAX 2048

__device__ __forceinline__ int klass(float x)
{
    const uint32_t u = __float_as_uint(x) & 0x7FFFFFFFu;
    if (u == 0u) return 3;
    if (u == 0x7F800000u) return 4;
    if (u > 0x7F800000u) return 5;
    const float a = fabsf(x);
    return a < 0x1p-90f ? 1 : (a > 0x1p90f ? 2 : 0);
}
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ __forceinline__ float rcp_nr(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float div3(float x, float y, float r)
{
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-y, q0, x);
    return __builtin_fmaf(e0, r, q0);
}
__device__ __forceinline__ float div5(float x, float y, float r)
{
    const float q1 = div3(x, y, r);
    const float e1 = __builtin_fmaf(-y, q1, x);
    return __builtin_fmaf(e1, r, q1);
}

__global__ void k_rcp(unsigned long long* bad, uint32_t* example)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (!same(rcp_nr(x), 1.0f / x)) { atomicAdd(&bad[klass(x)], 1ull); if (klass(x) == 0) atomicMax(example, (uint32_t)i & 0x7FFFFFFFu); }
    }
}
// blockIdx.y = res - 1
__global__ void k_div_res(unsigned long long* bad3, unsigned long long* bad5, uint32_t* ex3)
{
    const float y = (float)(blockIdx.y + 1u);
    const float r = 1.0f / y;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b3[6] = { 0, 0, 0, 0, 0, 0 }, b5[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float q = x / y;
        if (!same(div3(x, y, r), q)) { b3[klass(x)]++; if (klass(x) == 0) { ex3[0] = (uint32_t)i; ex3[1] = blockIdx.y + 1u; } }
        if (!same(div5(x, y, r), q)) b5[klass(x)]++;
    }
    for (int k = 0; k < 6; k++) { if (b3[k]) atomicAdd(&bad3[k], b3[k]); if (b5[k]) atomicAdd(&bad5[k], b5[k]); }
}
__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
__global__ void k_div_any(unsigned long long* bad3, unsigned long long* bad5, uint32_t* ex5)
{
    // divisor: sign +, exponent in [-24, 24), random mantissa; a 1/256 share of them with the mantissa all ones (the exception of the proofs)
    const uint32_t h = mix(blockIdx.y * 2654435761u + 12345u);
    uint32_t ym = h & 0x7FFFFFu;
    if ((blockIdx.y & 255u) == 255u) ym = 0x7FFFFFu;
    const uint32_t ye = 127u - 24u + (mix(h) % 48u);
    const float y = __uint_as_float((ye << 23) | ym);
    const float r = 1.0f / y;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 22); i += stride) {
        const float x = __uint_as_float(mix((uint32_t)i * 2246822519u + blockIdx.y));
        const float q = x / y;
        if (!same(div3(x, y, r), q)) atomicAdd(&bad3[klass(x)], 1ull);
        if (!same(div5(x, y, r), q)) { atomicAdd(&bad5[klass(x)], 1ull); if (klass(x) == 0) { ex5[0] = __float_as_uint(x); ex5[1] = __float_as_uint(y); } }
    }
}

static void show(const char* what, const unsigned long long* b)
{
    printf("%-44s mismatches: |x| in [2^-90, 2^90] %llu, smaller %llu, larger %llu, zero %llu, inf %llu, NaN %llu\n", what, b[0], b[1], b[2], b[3], b[4], b[5]);
}
int main()
{
    unsigned long long* d; uint32_t* ex;
    (void)hipMalloc(&d, 8 * 64); (void)hipMalloc(&ex, 64);
    (void)hipMemset(d, 0, 8 * 64); (void)hipMemset(ex, 0, 64);
    hipLaunchKernelGGL(k_rcp, dim3(4096), dim3(256), 0, 0, d, ex);
    hipLaunchKernelGGL(k_div_res, dim3(64, RES_MAX), dim3(256), 0, 0, d + 8, d + 16, ex + 2);
    hipLaunchKernelGGL(k_div_any, dim3(16, 1 << 14), dim3(256), 0, 0, d + 24, d + 32, ex + 4);
    (void)hipDeviceSynchronize();
    unsigned long long h[64]; uint32_t he[16];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(he, ex, sizeof he, hipMemcpyDeviceToHost);
    show("1/x by v_rcp_f32 + one Newton step, all x", h);
    printf("   largest |x| bits in the middle class that differ: 0x%08x\n", he[0]);
    char t[96];
    snprintf(t, sizeof t, "x / res, res = 1..%d, three operations", RES_MAX); show(t, h + 8);
    printf("   an example in the middle class: x bits 0x%08x, res %u\n", he[2], he[3]);
    snprintf(t, sizeof t, "x / res, res = 1..%d, five operations", RES_MAX); show(t, h + 16);
    show("x / y, 2^14 divisors x 2^22 x, three operations", h + 24);
    show("x / y, 2^14 divisors x 2^22 x, five operations", h + 32);
    printf("   an example in the middle class (five operations): x bits 0x%08x, y bits 0x%08x\n", he[4], he[5]);
    return 0;
}
