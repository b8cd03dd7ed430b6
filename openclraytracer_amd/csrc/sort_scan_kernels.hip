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

// ---- the levels inside a tile, eight elements per thread in registers ----
// A tile of 2^B elements (B <= 12) is held by 2^(B-3) threads, eight elements each: three bits of an element's index select the register
// (the "configuration" p: index bits p+2, p+1, p), the others the thread.  A compare-exchange level on one of the three bits is register
// arithmetic; to reach other bits the tile is transposed through LDS (eight 8-byte writes, a barrier, eight reads).  The merge of length
// 2^(m-1) -- levels on bits m-1 .. 0, in that order -- takes ceil(m / 3) configurations, the last one always (2, 1, 0), which is also how the
// tile is read from and written to memory (64 contiguous bytes per thread): 27 transposes sort 4 096 elements where one level per LDS round
// trip took 78 (44 -> see DESIGN.md, ClumsyDragon's 65 536 pairs).  Same comparators in the same order on every element as lds_levels
// (levels of one merge commute within a level only), `reverse` from the global index of the pair's lower element: the same permutation.
// LDS addresses are skewed by one element per eight (a thread's eight registers are 64 bytes apart in configuration 0).
#define BT_THREADS 512u
#define BT_PAD(e) ((e) + ((e) >> 3))
#define BT_LDS_ELEMS (SORT_TILE + (SORT_TILE >> 3))
__device__ __forceinline__ uint32_t bt_index(uint32_t t, uint32_t p, uint32_t j) { return ((t >> p) << (p + 3u)) | (t & ((1u << p) - 1u)) | (j << p); }

__device__ __forceinline__ void bt_transpose(uint2 (&x)[8], uint2* lds, uint32_t t, bool active, uint32_t from, uint32_t to)
{
    if (active) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) lds[BT_PAD(bt_index(t, from, j))] = x[j];
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) x[j] = lds[BT_PAD(bt_index(t, to, j))];
    }
    __syncthreads();
}

// the levels on the configuration's bits p + k for the k set in `mask` (descending), merge direction bit `dir`
__device__ __forceinline__ void bt_levels(uint2 (&x)[8], uint32_t t, uint32_t p, uint32_t mask, uint32_t dir, uint32_t tile_base)
{
    const uint32_t mine = tile_base + bt_index(t, p, 0u);
#pragma unroll
    for (int k = 2; k >= 0; k--) {
        if (!((mask >> k) & 1u)) continue;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            if (j & (1u << k)) continue;
            const bool reverse = ((dir & (mine | (j << p))) == 0u);
            order(x[j], x[j | (1u << k)], reverse);
        }
    }
}

// the merge with levels on bits m-1 .. 0 (direction bit `dir`); x is in configuration `cur` on entry and in configuration 0 on return
__device__ __forceinline__ void bt_merge(uint2 (&x)[8], uint2* lds, uint32_t t, bool active, uint32_t B, uint32_t m, uint32_t dir, uint32_t tile_base, uint32_t cur)
{
    int top = (int)m - 1;
    while (top >= 3) {
        const uint32_t nb = (uint32_t)(top - 2) < 3u ? (uint32_t)(top - 2) : 3u;       // bits top .. top - nb + 1, all above bit 2
        const uint32_t low = (uint32_t)top - nb + 1u;
        const uint32_t p = low < B - 3u ? low : B - 3u;
        if (cur != p) { bt_transpose(x, lds, t, active, cur, p); cur = p; }
        bt_levels(x, t, p, ((1u << nb) - 1u) << (low - p), dir, tile_base);
        top = (int)low - 1;
    }
    if (cur != 0u) bt_transpose(x, lds, t, active, cur, 0u);
    bt_levels(x, t, 0u, (1u << (top + 1)) - 1u, dir, tile_base);
}

// sorts every tile completely: lengths 1 .. tile_elems/2 (tile_elems = 2^B >= 8)
// (data2: a second array of the same size sorted by the same launch, blockIdx.y = 1 -- a BVH build sorts the keys of its two hierarchies, and a
//  launch costs its ~5 us whatever it carries)
__global__ __launch_bounds__(BT_THREADS) void k_bitonic_tile_sort(uint2* data, uint2* data2, uint32_t B)
{
    __shared__ uint2 lds[BT_LDS_ELEMS];
    if (blockIdx.y) data = data2;
    const uint32_t t = threadIdx.x, base = blockIdx.x << B;
    const bool active = t < (1u << (B - 3u));
    uint2 x[8];
    if (active) {
        const uint4* src = reinterpret_cast<const uint4*>(data + base + 8u * t);
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) { const uint4 v = src[q]; x[2 * q] = make_uint2(v.x, v.y); x[2 * q + 1] = make_uint2(v.z, v.w); }
    }
    for (uint32_t m = 1; m <= B; m++) bt_merge(x, lds, t, active, B, m, 1u << m, base, 0u);
    if (active) {
        uint4* dst = reinterpret_cast<uint4*>(data + base + 8u * t);
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) dst[q] = make_uint4(x[2 * q].x, x[2 * q].y, x[2 * q + 1].x, x[2 * q + 1].y);
    }
}

// the tail (levels on bits B-1 .. 0) of the merge with the given length >= 2^B
__global__ __launch_bounds__(BT_THREADS) void k_bitonic_tile_merge(uint2* data, uint2* data2, uint32_t B, uint32_t dir)
{
    __shared__ uint2 lds[BT_LDS_ELEMS];
    if (blockIdx.y) data = data2;
    const uint32_t t = threadIdx.x, base = blockIdx.x << B;
    const bool active = t < (1u << (B - 3u));
    uint2 x[8];
    const uint32_t p0 = B - 3u;                                   // read in the configuration of the first three levels: consecutive lanes, consecutive elements
    if (active) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) x[j] = data[base + bt_index(t, p0, j)];
    }
    bt_merge(x, lds, t, active, B, B, dir, base, p0);
    if (active) {
        uint4* dst = reinterpret_cast<uint4*>(data + base + 8u * t);
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) dst[q] = make_uint4(x[2 * q].x, x[2 * q].y, x[2 * q + 1].x, x[2 * q + 1].y);
    }
}

// (tiles of fewer than eight elements -- N = 2, 4 -- keep one level per LDS round trip)
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_lds_sort(uint2* data, uint2* data2, uint32_t tile_elems)
{
    __shared__ uint2 tile[SORT_TILE];
    if (blockIdx.y) data = data2;
    uint32_t base = blockIdx.x * tile_elems;
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) tile[t] = data[base + t];
    __syncthreads();
    for (uint32_t length = 1; length < tile_elems; length <<= 1)
        lds_levels(tile, tile_elems, base, length, length << 1);
    for (uint32_t t = threadIdx.x; t < tile_elems; t += blockDim.x) data[base + t] = tile[t];
}

// Up to 8 consecutive compare-exchange levels inc = 2^top .. 2^(top-levels+1), all >= SORT_TILE, in ONE pass through HBM.
// (The reference fuses two or three levels per launch in registers, ParallelBitonic_B4 / _B8, SortKernels.cl:290-345, selection
// BitonicSort.cpp:127-153; one level per launch made the networks over the grid's 1-2 M pairs launch bound.)
// Elements that meet in these levels differ only in index bits [low, top], low = top - levels + 1: a block stages
// R = 2^levels rows (one per combination of those bits) of C = SORT_TILE / R consecutive elements in LDS -- runs of C * 8
// bytes (>= 128) in HBM -- and runs the levels there, row against row.  Same comparators, same `reverse` per pair (taken from
// the global index of the pair's first element), hence the same permutation as one level per launch.
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_strided(uint2* data, uint2* data2, uint32_t top, uint32_t levels, uint32_t dir)
{
    __shared__ uint2 tile[SORT_TILE];
    if (blockIdx.y) data = data2;
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

void launch_bitonic_sort(hipStream_t s, uint2* pairs, uint32_t N, uint2* pairs2)
{
    if (N < 2) return;
    uint32_t tile = N < SORT_TILE ? N : SORT_TILE;
    uint32_t blocks = N / tile;
    const uint32_t B = log2u(tile), arrays = pairs2 ? 2u : 1u;
    if (tile < 8u) hipLaunchKernelGGL(k_bitonic_lds_sort, dim3(blocks, arrays), dim3(SORT_THREADS), 0, s, pairs, pairs2, tile);
    else hipLaunchKernelGGL(k_bitonic_tile_sort, dim3(blocks, arrays), dim3(BT_THREADS), 0, s, pairs, pairs2, B);
    for (uint32_t length = tile; length < N; length <<= 1) {
        // levels inc = length .. tile go through HBM, up to eight per pass; the rest of the merge runs inside the tiles
        uint32_t top = log2u(length);
        uint32_t left = top - 12u + 1u;                     // tile == SORT_TILE == 2^12 here (N > tile)
        while (left) {
            const uint32_t levels = left < SORT_FUSED_LEVELS ? left : SORT_FUSED_LEVELS;
            hipLaunchKernelGGL(k_bitonic_strided, dim3(N / SORT_TILE, arrays), dim3(SORT_THREADS), 0, s, pairs, pairs2, top, levels, length << 1);
            top -= levels; left -= levels;
        }
        hipLaunchKernelGGL(k_bitonic_tile_merge, dim3(blocks, arrays), dim3(BT_THREADS), 0, s, pairs, pairs2, B, length << 1);
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

// ----------------------------------------------------------------------------------------
// LSD radix sort of uint2 (key = .x), 8 bits per pass, STABLE: equal keys keep their input order (the bitonic network above
// permutes them the way the reference's network does, which is part of the reference's result; this sort is for orders that are
// the library's own -- the search tree's keys, ray bins -- and for the FAST build mode).  HBM-bound integer work: per pass the keys
// are read twice and the pairs written once (n x 24 B), nothing is reshaped into anything else.
//   k_radix_hist     one block per tile of RDX_TILE elements: 256-bin digit histogram (LDS atomics) -> hist[digit][tile]
//   k_radix_rows     one block per digit: exclusive scan of its row of tile counts (coalesced), row total -> totals[digit]
//   k_radix_scatter  one block per tile: every wave ranks its 1 024 consecutive elements digit by digit -- lanes with the same digit
//                    find each other with eight ballots, the lowest of them bumps the wave's counter of that digit in LDS, a lane's
//                    rank is the old count plus the equal-digit lanes below it -- then the counts of the four waves, the tile's
//                    digit starts and the global starts (row prefix + scan of the totals) are combined, the pairs are staged in LDS
//                    in sorted order and written out in runs of consecutive addresses per digit
//   k_radix_small    n <= RDX_TILE: all passes in one block, in LDS
// ----------------------------------------------------------------------------------------
#define RDX_BINS 256u
#define RDX_THREADS 256u
#define RDX_WAVES (RDX_THREADS / 64u)
// elements per thread R (template argument): a tile is R x 256 elements, a wave's share R x 64 consecutive ones in R rounds of 64.
// R = 16 (tiles of 4 096: 32 KB of staging, long runs per digit in the scatter) for large inputs; R = 4 (tiles of 1 024) below
// RDX_SMALL_TILES_BELOW elements, where 4 096-element tiles would leave most of the 256 CUs idle and every block's 16 dependent rounds
// are the pass's latency (50 032 pairs: 13 blocks, 10.8 us per scatter pass)
#define RDX_SMALL_TILES_BELOW (1u << 19)

__device__ __forceinline__ uint32_t rdx_lanes_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// block-wide exclusive scan of one value per thread (256 threads); *total = the sum
__device__ __forceinline__ uint32_t rdx_block_exclusive(uint32_t v, uint32_t* wave_sums /* RDX_WAVES words of LDS */, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= (uint32_t)off) inc += o; }
    if (lane == 63u) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t prefix = 0, sum = 0;
    for (uint32_t w = 0; w < RDX_WAVES; w++) { const uint32_t ws = wave_sums[w]; if (w < wave) prefix += ws; sum += ws; }
    __syncthreads();
    *total = sum;
    return inc - v + prefix;
}

// (in2 / hist2 / ...: a second sort of the same size carried by the same launches, blockIdx.y = 1)
template <uint32_t R>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_hist(const uint2* __restrict__ in, uint32_t n, uint32_t shift, uint32_t tiles, uint32_t* __restrict__ hist,
                                                            const uint2* __restrict__ in2, uint32_t* __restrict__ hist2)
{
    __shared__ uint32_t cnt[RDX_BINS];
    if (blockIdx.y) { in = in2; hist = hist2; }
    cnt[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (R * RDX_THREADS);
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t i = base + k * RDX_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(in[i].x >> shift) & (RDX_BINS - 1u)], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * tiles + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(RDX_THREADS) void k_radix_rows(uint32_t* __restrict__ hist, uint32_t tiles, uint32_t* __restrict__ totals,
                                                            uint32_t* __restrict__ hist2, uint32_t* __restrict__ totals2)
{
    __shared__ uint32_t wave_sums[RDX_WAVES];
    if (blockIdx.y) { hist = hist2; totals = totals2; }
    uint32_t* row = hist + (size_t)blockIdx.x * tiles;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < tiles; t0 += RDX_THREADS) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < tiles ? row[t] : 0u;
        uint32_t total;
        const uint32_t ex = rdx_block_exclusive(v, wave_sums, &total);
        if (t < tiles) row[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// ranks of a wave's elements among the wave's elements with the same digit seen so far; cnt = this wave's 256 counters in LDS
__device__ __forceinline__ uint32_t rdx_rank(uint32_t digit, bool live, uint32_t* cnt)
{
    uint64_t m = __ballot(live);
#pragma unroll
    for (uint32_t b = 0; b < 8u; b++) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t bal = __ballot(live && bit);
        m &= bit ? bal : ~bal;
    }
    // (a lane that is not live holds garbage in m: it neither counts nor is counted)
    const uint32_t below = rdx_lanes_below(m);
    uint32_t old = 0;
    if (live && below == 0u) { old = cnt[digit]; cnt[digit] = old + (uint32_t)__popcll(m); }
    const int leader = live ? (int)__builtin_ctzll(m) : 0;
    old = (uint32_t)__shfl((int)old, leader, 64);
    return old + below;
}

// one pass over the block's R x 256 elements held in registers (element k of a thread: index wave * R * 64 + k * 64 + lane):
// ranks them, stages them in `stage` in sorted order, leaves the tile's digit starts in dig_start.  count = live elements.
template <uint32_t R>
__device__ __forceinline__ void rdx_tile_pass(const uint2 (&e)[R], uint32_t count, uint32_t shift, uint32_t (*cnt)[RDX_BINS],
                                              uint32_t* dig_start, uint32_t* wave_sums, uint2* stage)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t w = 0; w < RDX_WAVES; w++) cnt[w][threadIdx.x] = 0u;
    __syncthreads();
    uint32_t rank[R];
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
        rank[k] = rdx_rank((e[k].x >> shift) & (RDX_BINS - 1u), idx < count, cnt[wave]);
    }
    __syncthreads();
    {   // thread d: digit d's counts of the four waves -> their exclusive prefix; the tile's digit starts
        uint32_t run = 0;
        for (uint32_t w = 0; w < RDX_WAVES; w++) { const uint32_t c = cnt[w][threadIdx.x]; cnt[w][threadIdx.x] = run; run += c; }
        uint32_t total;
        dig_start[threadIdx.x] = rdx_block_exclusive(run, wave_sums, &total);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
        if (idx < count) {
            const uint32_t d = (e[k].x >> shift) & (RDX_BINS - 1u);
            stage[dig_start[d] + cnt[wave][d] + rank[k]] = e[k];
        }
    }
    __syncthreads();
}

template <uint32_t R>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_scatter(const uint2* __restrict__ in, uint2* __restrict__ out, uint32_t n, uint32_t shift,
                                                               uint32_t tiles, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                                                               const uint2* __restrict__ in2, uint2* __restrict__ out2, const uint32_t* __restrict__ hist2,
                                                               const uint32_t* __restrict__ totals2)
{
    __shared__ uint32_t cnt[RDX_WAVES][RDX_BINS];
    __shared__ uint32_t dig_start[RDX_BINS];
    __shared__ uint32_t gbase[RDX_BINS];
    if (blockIdx.y) { in = in2; out = out2; hist = hist2; totals = totals2; }
    __shared__ uint32_t wave_sums[RDX_WAVES];
    __shared__ uint2 stage[R * RDX_THREADS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * (R * RDX_THREADS);
    const uint32_t count = n - base < R * RDX_THREADS ? n - base : R * RDX_THREADS;
    uint2 e[R];
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
        e[k] = idx < count ? in[base + idx] : make_uint2(0u, 0u);
    }
    {   // where this tile's run of digit d starts in the output: elements with smaller digits anywhere + this digit in earlier tiles
        uint32_t total;
        const uint32_t before = rdx_block_exclusive(totals[threadIdx.x], wave_sums, &total);
        gbase[threadIdx.x] = before + hist[threadIdx.x * tiles + blockIdx.x];
    }
    rdx_tile_pass<R>(e, count, shift, cnt, dig_start, wave_sums, stage);
    for (uint32_t i = threadIdx.x; i < count; i += RDX_THREADS) {
        const uint2 v = stage[i];
        const uint32_t d = (v.x >> shift) & (RDX_BINS - 1u);
        out[gbase[d] + (i - dig_start[d])] = v;
    }
}

#define RDX_ONE_BLOCK 4096u       // up to here: all passes in one block, in LDS (R = 16; R = 4 up to 1 024 and R = 1 up to 256 elements: a
                                  // dozen pairs do not pay for sixteen rounds of ballots per pass -- 21.6 -> 4 us on the cube's 12)
template <uint32_t R>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_small(uint2* __restrict__ data, uint32_t n, uint32_t passes, uint2* __restrict__ data2)
{
    __shared__ uint32_t cnt[RDX_WAVES][RDX_BINS];
    if (blockIdx.y) data = data2;
    __shared__ uint32_t dig_start[RDX_BINS];
    __shared__ uint32_t wave_sums[RDX_WAVES];
    __shared__ uint2 stage[R * RDX_THREADS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 e[R];
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
        e[k] = idx < n ? data[idx] : make_uint2(0u, 0u);
    }
    for (uint32_t p = 0; p < passes; p++) {
        rdx_tile_pass<R>(e, n, p * 8u, cnt, dig_start, wave_sums, stage);
#pragma unroll
        for (uint32_t k = 0; k < R; k++) {
            const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
            if (idx < n) e[k] = stage[idx];
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t idx = wave * (R * 64u) + k * 64u + lane;
        if (idx < n) data[idx] = e[k];
    }
}

// (Round 4 tried to shorten the chain of a build-sized sort -- twelve launches of ~4-6 us -- with 10-bit digits and ONE launch per pass: the tile
//  histograms of the next digit counted by the scatter itself, first with global atomics into the row of the tile an element lands in (a wave's 64
//  elements land in one tile and share few next digits: 64 adds to two or three lines per instruction, 30 us per pass at 50 032 pairs), then as rows
//  each tile publishes for the tiles behind it (count | 2^31 per word, device-scope stores and polls, tiles taken from a ticket).  Correct, and
//  slower than the three small kernels: a block that sums 49 rows of 4 KB written by other XCDs a moment ago waits ~2.5 us per batch of sixteen
//  loads, 10 us per pass for the digit totals and 10 more for the rows before it, on top of 8 us for the 1 024-bin bookkeeping alone -- 20-29 us
//  per pass against 13.  Removed; the measurements are in DESIGN.md section 4.)
static uint32_t radix_tile(uint32_t n) { return (n < RDX_SMALL_TILES_BELOW ? 4u : 16u) * RDX_THREADS; }
// scratch words a sort of n pairs needs behind its n-pair ping-pong buffer: hist[256][tiles] + totals[256]
size_t radix_sort_scratch_words(uint32_t n) { return (size_t)RDX_BINS * ((n + 1023u) / 1024u) + RDX_BINS; }

// sorts pairs[0, n) by the low key_bits bits of .x (the bits above must be equal on all keys, e.g. zero), stable; tmp holds n pairs,
// scratch radix_sort_scratch_words(n) words.  The result is in `pairs`.  pairs2 / tmp2 / scratch2: a second sort of the same size and key
// width carried by the same launches.
void launch_radix_sort(hipStream_t s, uint2* pairs, uint2* tmp, uint32_t n, uint32_t key_bits, uint32_t* scratch, uint2* pairs2, uint2* tmp2, uint32_t* scratch2)
{
    if (n < 2) return;
    const uint32_t passes = key_bits == 0 ? 1u : (key_bits + 7u) / 8u;
    const uint32_t arrays = pairs2 ? 2u : 1u;
    if (n <= RDX_ONE_BLOCK) {
        if (n <= RDX_THREADS) hipLaunchKernelGGL(k_radix_small<1u>, dim3(1, arrays), dim3(RDX_THREADS), 0, s, pairs, n, passes, pairs2);
        else if (n <= 4u * RDX_THREADS) hipLaunchKernelGGL(k_radix_small<4u>, dim3(1, arrays), dim3(RDX_THREADS), 0, s, pairs, n, passes, pairs2);
        else hipLaunchKernelGGL(k_radix_small<16u>, dim3(1, arrays), dim3(RDX_THREADS), 0, s, pairs, n, passes, pairs2);
        return;
    }
    const uint32_t tile = radix_tile(n), tiles = (n + tile - 1u) / tile;
    uint32_t* hist = scratch; uint32_t* totals = scratch + (size_t)RDX_BINS * tiles;
    uint32_t* hist2 = scratch2; uint32_t* totals2 = scratch2 ? scratch2 + (size_t)RDX_BINS * tiles : nullptr;
    uint2* src = pairs; uint2* dst = tmp;
    uint2* src2 = pairs2; uint2* dst2 = tmp2;
    for (uint32_t p = 0; p < passes; p++) {
        if (tile == 4u * RDX_THREADS) hipLaunchKernelGGL(k_radix_hist<4u>, dim3(tiles, arrays), dim3(RDX_THREADS), 0, s, (const uint2*)src, n, p * 8u, tiles, hist, (const uint2*)src2, hist2);
        else hipLaunchKernelGGL(k_radix_hist<16u>, dim3(tiles, arrays), dim3(RDX_THREADS), 0, s, (const uint2*)src, n, p * 8u, tiles, hist, (const uint2*)src2, hist2);
        hipLaunchKernelGGL(k_radix_rows, dim3(RDX_BINS, arrays), dim3(RDX_THREADS), 0, s, hist, tiles, totals, hist2, totals2);
        if (tile == 4u * RDX_THREADS) hipLaunchKernelGGL(k_radix_scatter<4u>, dim3(tiles, arrays), dim3(RDX_THREADS), 0, s, (const uint2*)src, dst, n, p * 8u, tiles, (const uint32_t*)hist, (const uint32_t*)totals,
                                                         (const uint2*)src2, dst2, (const uint32_t*)hist2, (const uint32_t*)totals2);
        else hipLaunchKernelGGL(k_radix_scatter<16u>, dim3(tiles, arrays), dim3(RDX_THREADS), 0, s, (const uint2*)src, dst, n, p * 8u, tiles, (const uint32_t*)hist, (const uint32_t*)totals,
                                (const uint2*)src2, dst2, (const uint32_t*)hist2, (const uint32_t*)totals2);
        uint2* t = src; src = dst; dst = t;
        t = src2; src2 = dst2; dst2 = t;
    }
    if (src != pairs) {
        (void)hipMemcpyAsync(pairs, src, (size_t)n * sizeof(uint2), hipMemcpyDeviceToDevice, s);
        if (pairs2) (void)hipMemcpyAsync(pairs2, src2, (size_t)n * sizeof(uint2), hipMemcpyDeviceToDevice, s);
    }
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
