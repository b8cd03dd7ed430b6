// sort_scan_kernels.hip -- LDS-staged parallel primitives for the builders (gfx950).
//
//  * bitonic sort of uint2 (key = .x): the SAME comparator network as the reference
//    (BitonicSort.cpp:118-189 driving ParallelBitonic_B2/B4/B8, SortKernels.cl:266-345), so equal
//    keys end up in the same order -- that order decides the radix-tree topology, the DFS order
//    and therefore which of two equal-t hits wins.  The reference runs one launch (+flush+wait)
//    per 1-3 compare-exchange levels, all through global memory; here every level with
//    inc < TILE runs inside LDS and only the levels with inc >= TILE touch HBM.
//  * inclusive uint scan (PrefixSumKernels.cl:28-106 semantics: out[i] = sum_{j<=i} in[j]).
#include "rtmi_internal.h"

#define SORT_TILE 4096u      // elements per block: 32 KiB of LDS
#define SORT_THREADS 512u

// ORDER(a,b), SortKernels.cl:266: swap = reverse ^ (key(a) < key(b))
__device__ __forceinline__ void order(uint2& a, uint2& b, bool reverse)
{
    bool swap = reverse ^ (a.x < b.x);
    uint2 auxa = a, auxb = b;
    a = swap ? auxb : auxa;
    b = swap ? auxa : auxb;
}

// all compare-exchange levels inc = inc_start .. 1 of merge `dir` inside one LDS tile
__device__ __forceinline__ void lds_levels(uint2* tile, uint32_t tile_elems, uint32_t tile_base, uint32_t inc_start,
                                           uint32_t dir)
{
    for (uint32_t inc = inc_start; inc > 0; inc >>= 1) {
        for (uint32_t t = threadIdx.x; t < tile_elems / 2; t += blockDim.x) {
            uint32_t low = t & (inc - 1);
            uint32_t i = (t << 1) - low;                       // insert 0 at bit position of inc
            bool reverse = ((dir & (tile_base + i)) == 0);
            uint2 x0 = tile[i], x1 = tile[i + inc];
            order(x0, x1, reverse);
            tile[i] = x0; tile[i + inc] = x1;
        }
        __syncthreads();
    }
}

// sorts every tile completely: lengths 1 .. tile_elems/2
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_lds_sort(uint2* data, uint32_t tile_elems)
{
    __shared__ uint2 tile[SORT_TILE];
    uint32_t base = blockIdx.x * tile_elems;
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) tile[t] = data[base + t];
    __syncthreads();
    for (uint32_t length = 1; length < tile_elems; length <<= 1)
        lds_levels(tile, tile_elems, base, length, length << 1);
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) data[base + t] = tile[t];
}

// the tail (inc = tile/2 .. 1) of the merge with the given length >= tile_elems
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_lds_merge(uint2* data, uint32_t tile_elems, uint32_t length)
{
    __shared__ uint2 tile[SORT_TILE];
    uint32_t base = blockIdx.x * tile_elems;
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) tile[t] = data[base + t];
    __syncthreads();
    lds_levels(tile, tile_elems, base, tile_elems >> 1, length << 1);
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) data[base + t] = tile[t];
}

// one compare-exchange level through HBM (ParallelBitonic_B2, SortKernels.cl:269-287)
__global__ __launch_bounds__(256) void k_bitonic_global(uint2* data, uint32_t half_n, uint32_t inc, uint32_t dir)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= half_n) return;
    uint32_t low = t & (inc - 1);
    uint32_t i = (t << 1) - low;
    bool reverse = ((dir & i) == 0);
    uint2 x0 = data[i], x1 = data[i + inc];
    order(x0, x1, reverse);
    data[i] = x0; data[i + inc] = x1;
}

void launch_bitonic_sort(hipStream_t s, uint2* pairs, uint32_t N)
{
    if (N < 2) return;
    uint32_t tile = N < SORT_TILE ? N : SORT_TILE;
    uint32_t blocks = N / tile;
    hipLaunchKernelGGL(k_bitonic_lds_sort, dim3(blocks), dim3(SORT_THREADS), 0, s, pairs, tile);
    for (uint32_t length = tile; length < N; length <<= 1) {
        for (uint32_t inc = length; inc >= tile; inc >>= 1)
            hipLaunchKernelGGL(k_bitonic_global, dim3((N / 2 + 255) / 256), dim3(256), 0, s, pairs, N / 2, inc,
                               length << 1);
        hipLaunchKernelGGL(k_bitonic_lds_merge, dim3(blocks), dim3(SORT_THREADS), 0, s, pairs, tile, length);
    }
}

// ----------------------------------------------------------------------------------------
// inclusive scan, three phases: per-block totals -> scan of totals (recursive) -> per-block
// scan seeded with the exclusive block prefix.  2048 elements per 256-thread block.
// ----------------------------------------------------------------------------------------
#define SCAN_THREADS 256u
#define SCAN_PER_THREAD 8u
#define SCAN_BLOCK (SCAN_THREADS * SCAN_PER_THREAD)

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    unsigned lane = threadIdx.x & 63u;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= (unsigned)off) v += o;
    }
    return v;
}

// block-wide inclusive scan of one value per thread; returns the block total through *total
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t prefix = 0, sum = 0;
    for (unsigned w = 0; w < SCAN_THREADS / 64; w++) {
        uint32_t ws = wave_sums[w];
        if (w < wave) prefix += ws;
        sum += ws;
    }
    __syncthreads();
    *total = sum;
    return inc + prefix;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_totals(const uint32_t* in, uint32_t* totals, uint32_t n)
{
    uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < SCAN_PER_THREAD; k++)
        if (base + k < n) acc += in[base + k];
    uint32_t total;
    block_inclusive_scan(acc, &total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocks(const uint32_t* in, uint32_t* out, uint32_t n,
                                                              const uint32_t* block_prefix_inclusive)
{
    uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    uint32_t acc = 0;
    for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        acc += v[k];
    }
    uint32_t total;
    uint32_t inc = block_inclusive_scan(acc, &total);
    uint32_t run = inc - acc;                                    // exclusive prefix of this thread
    if (block_prefix_inclusive && blockIdx.x > 0) run += block_prefix_inclusive[blockIdx.x - 1];
    for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
}

size_t scan_scratch_words(uint32_t n)
{
    size_t words = 0;
    while (n > SCAN_BLOCK) {
        uint32_t blocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
        words += blocks;
        n = blocks;
    }
    return words + 1;
}

int launch_scan_inclusive(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch)
{
    if (n == 0) return 0;
    uint32_t blocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (blocks == 1) {
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, n, (const uint32_t*)nullptr);
        return 0;
    }
    uint32_t* totals = scratch;
    hipLaunchKernelGGL(k_scan_block_totals, dim3(blocks), dim3(SCAN_THREADS), 0, s, in, totals, n);
    launch_scan_inclusive(s, totals, totals, blocks, scratch + blocks);        // in-place scan of the totals
    hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(SCAN_THREADS), 0, s, in, out, n, (const uint32_t*)totals);
    return 0;
}
