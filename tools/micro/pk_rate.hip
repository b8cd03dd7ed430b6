// pk_rate.hip -- issue cost of packed f32 VALU instructions next to their scalar forms on gfx950, one wave per SIMD and eight:
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_rate tools/micro/pk_rate.hip && tools/micro/pk_rate
// Each variant runs ITER iterations of 16 independent instructions of one kind (inline asm, so the compiler neither packs nor unpacks them);
// reported: shader cycles per instruction per wave (s_memtime), and per f32 result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float seed)
{
    float a[16]; float2v p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + (float)i + threadIdx.x; p[i] = float2v{ a[i], a[i] + 0.5f }; }
    const float m = 1.0000001f; const float2v mp = { m, m };
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_mov_b32 s20, 0x33333333\n\ts_mov_b32 s21, 0x33333333" : : : "vcc", "s20", "s21");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mp));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mp));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(mp));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 8) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 9) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 12) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            if (KIND == 13) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m));
            if (KIND == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            if (KIND == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 17) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 18) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 20) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 21) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 22) asm volatile("v_cmp_lt_f32 s[22:23], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[i]) : "v"(m) : "s22", "s23");
            if (KIND == 23) asm volatile("v_cmp_lt_f32 s[22:23], %0, %1" : : "v"(a[i]), "v"(m) : "s22", "s23");
            if (KIND == 24) asm volatile("v_cmpx_lt_f32 exec, %0, %0\n\ts_mov_b64 exec, -1" : : "v"(a[i]) : "exec");
            if (KIND == 25) asm volatile("s_and_saveexec_b64 s[22:23], vcc\n\ts_mov_b64 exec, s[22:23]" : : : "s22", "s23", "exec");
            if (KIND == 26) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 27) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int results_per_instr)
{
    float* out; unsigned long long* cyc;
    const int blocks = 256 * 4;
    hipMalloc(&out, sizeof(float) * blocks * 512); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int waves_per_simd : { 1, 2, 8 }) {
        const int threads = 64 * waves_per_simd;            // one block per SIMD's worth: 4 blocks per CU
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const double per = (double)h[0] / (ITER * 16.0);
        printf("%-18s block of %d wave(s): %7.2f counter ticks per instruction per wave, kernel %.3f ms -> %.3f ns per instruction per wave, %.3f per f32 result\n",
               name, waves_per_simd, per, ms, ms * 1e6 / (ITER * 16.0), ms * 1e6 / (ITER * 16.0) / results_per_instr);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_mul_f32", 1); run<1>("v_pk_mul_f32", 2); run<2>("v_add_f32", 1); run<3>("v_pk_add_f32", 2);
    run<4>("v_fma_f32", 1); run<5>("v_pk_fma_f32", 2); run<6>("v_rcp_f32", 1); run<7>("v_div_scale_f32", 1);
    run<8>("v_div_fmas_f32", 1); run<9>("v_div_fixup_f32", 1); run<10>("v_cndmask_b32", 1); run<11>("v_mov_b32", 1);
    run<12>("v_cvt_f32_u32", 1); run<13>("v_min3_f32", 1);
    run<14>("v_cndmask_e64 sgpr", 1); run<15>("v_cmp_lt_f32 vcc", 1); run<16>("v_cmp+v_cndmask vcc", 1); run<17>("v_max_f32", 1);
    run<18>("v_cvt_i32_f32", 1); run<19>("v_mul_lo_u32", 1); run<20>("v_lshl_add_u32", 1); run<21>("v_add_u32", 1);
    run<22>("v_cmp+v_cndmask sgpr", 1); run<23>("v_cmp_lt_f32 sgpr", 1); run<24>("v_cmpx + s_mov exec", 1); run<25>("s_and_saveexec + s_mov", 1);
    run<26>("v_sub_f32", 1); run<27>("v_med3_f32", 1);
    return 0;
}
