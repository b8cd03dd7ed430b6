// sort_scan_kernels.hip -- LDS-staged parallel primitives for the builders (gfx950).
//
//  * bitonic sort of uint2 (key = .x): the SAME comparator network as the reference
//    (BitonicSort.cpp:118-189 driving ParallelBitonic_B2/B4/B8, SortKernels.cl:266-345), so equal
//    keys end up in the same order -- that order decides the radix-tree topology, the DFS order
//    and therefore which of two equal-t hits wins.  The reference runs one launch (+flush+wait)
//    per 1-3 compare-exchange levels, all through global memory; here every level with
//    inc < TILE runs inside LDS and the levels with inc >= TILE go through HBM up to eight at a time.
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

// Up to 8 consecutive compare-exchange levels inc = 2^top .. 2^(top-levels+1), all >= SORT_TILE, in ONE pass through HBM.
// (The reference fuses two or three levels per launch in registers, ParallelBitonic_B4 / _B8, SortKernels.cl:290-345, selection
// BitonicSort.cpp:127-153; one level per launch made the networks over the grid's 1-2 M pairs launch bound.)
// Elements that meet in these levels differ only in index bits [low, top], low = top - levels + 1: a block stages
// R = 2^levels rows (one per combination of those bits) of C = SORT_TILE / R consecutive elements in LDS -- runs of C * 8
// bytes (>= 128) in HBM -- and runs the levels there, row against row.  Same comparators, same `reverse` per pair (taken from
// the global index of the pair's first element), hence the same permutation as one level per launch.
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_strided(uint2* data, uint32_t top, uint32_t levels, uint32_t dir)
{
    __shared__ uint2 tile[SORT_TILE];
    const uint32_t low = top + 1u - levels;                 // lowest varying bit; low >= log2(SORT_TILE) >= log2(C)
    const uint32_t rows = 1u << levels, cols = SORT_TILE >> levels;
    const uint32_t cols_log = 12u - levels;                 // log2(SORT_TILE) = 12
    const uint32_t chunks_log = low - cols_log;             // runs of `cols` elements below bit `low`
    const uint32_t chunk = blockIdx.x & ((1u << chunks_log) - 1u), high = blockIdx.x >> chunks_log;
    const uint32_t base = (high << (top + 1u)) | (chunk << cols_log);
    for (uint32_t t = threadIdx.x; t < SORT_TILE; t += SORT_THREADS) {
        const uint32_t g = t >> cols_log, c = t & (cols - 1u);
        tile[t] = data[base + (g << low) + c];
    }
    __syncthreads();
    for (int j = (int)levels - 1; j >= 0; j--) {
        for (uint32_t q = threadIdx.x; q < SORT_TILE / 2u; q += SORT_THREADS) {
            const uint32_t c = q & (cols - 1u), gp = q >> cols_log;             // gp: the row index without bit j
            const uint32_t g = ((gp >> j) << (j + 1)) | (gp & ((1u << j) - 1u));
            const uint32_t a = (g << cols_log) + c, b = a + ((1u << j) << cols_log);
            const bool reverse = ((dir & (base + (g << low) + c)) == 0u);
            uint2 x0 = tile[a], x1 = tile[b];
            order(x0, x1, reverse);
            tile[a] = x0; tile[b] = x1;
        }
        __syncthreads();
    }
    for (uint32_t t = threadIdx.x; t < SORT_TILE; t += SORT_THREADS) {
        const uint32_t g = t >> cols_log, c = t & (cols - 1u);
        data[base + (g << low) + c] = tile[t];
    }
    (void)rows;
}

#define SORT_FUSED_LEVELS 8u

static uint32_t log2u(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) l++; return l; }

void launch_bitonic_sort(hipStream_t s, uint2* pairs, uint32_t N)
{
    if (N < 2) return;
    uint32_t tile = N < SORT_TILE ? N : SORT_TILE;
    uint32_t blocks = N / tile;
    hipLaunchKernelGGL(k_bitonic_lds_sort, dim3(blocks), dim3(SORT_THREADS), 0, s, pairs, tile);
    for (uint32_t length = tile; length < N; length <<= 1) {
        // levels inc = length .. tile go through HBM, up to eight per pass; the rest of the merge runs in LDS
        uint32_t top = log2u(length);
        uint32_t left = top - 12u + 1u;                     // tile == SORT_TILE == 2^12 here (N > tile)
        while (left) {
            const uint32_t levels = left < SORT_FUSED_LEVELS ? left : SORT_FUSED_LEVELS;
            hipLaunchKernelGGL(k_bitonic_strided, dim3(N / SORT_TILE), dim3(SORT_THREADS), 0, s, pairs, top, levels, length << 1);
            top -= levels; left -= levels;
        }
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

// rt_ctx_create loads this file's code object up front (HIP loads a code object on the first use of one of its kernels: the first
// frame of a process otherwise pays for it -- the traversal kernels' object alone takes ~2 ms)
bool rt_preload_sort_scan()
{
    hipFuncAttributes attr;
    const bool ok = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_bitonic_lds_sort)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
