// radix_sort.hip -- hand-written stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 (wave64).
//
// Used once per frame: order the visible Gaussians by their 32-bit depth key (4 passes x 8 bits; the value of the first pass is the
// implicit index, the culled Gaussians are dropped by it: `n_live` below), and by the kNN grid for its Morton codes (knn.hip).
// Stability is what makes the final list equal to the reference's 64-bit (tile | depth) sort: the tile partition that follows
// (binning.hip) keeps the depth order it is handed.
//
// One pass = histogram -> one row scan (a block per digit) -> scatter (every scatter block turns the <= 256 row totals into digit
// bases itself: no third tiny kernel in the dependency chain).  A 256-thread block owns 2048 consecutive items; each
// wave owns 512 of them and ranks them 64 at a time with one LDS atomic per item on the (wave, digit) run counter (see the rank
// phase below).  No atomics on global memory, integer work only, no MFMA.
#include <cstdlib>
#include "common.h"

namespace sr {

constexpr int kRsThreads = 256;
constexpr int kRsItems = 8;                       // per thread (the emission-offset scan and the rank test hook)
constexpr int kRsTile = kRsThreads * kRsItems;    // 2048 items per block
static_assert(kRsTile == kScanTile, "first_index() divides by the scan's block size");
// The sort's histogram / scatter kernels take 8 items per thread (2048 per block), or 16 for 8-bit passes over >= 2 M items: half the
// blocks, digit runs twice as long in the write-out (depth sort at 3 M keys 0.200 -> 0.189 ms; below ~1 M items the 2048-item
// blocks fill the chip better).
constexpr int kSortItemsBig = 16;
constexpr uint32_t kSortBigFrom = 2u << 20;
constexpr int kRsMaxBins = 256;
// The payload rides along with the value from this many items up (below, the whole payload table sits in the L2 and the last pass's
// gather is the cheaper way: 500 k items 69 us gathered / 73 us riding; 3 M: 181 / 167; 6 M: 334 / 295)
constexpr uint32_t kSortRideFrom = 1u << 20;

// Compacting sorts (the depth sort: `n_dev` = the frame's visible count, emission scan): the FIRST pass reads all n_host items and drops
// those whose key is kDropKey (culled Gaussians) -- they are neither counted nor written -- and every later pass works on the *n_dev
// survivors only: its blocks beyond them leave at once and the row scan stops where they stop.  A frame seen from inside the cloud (a
// street: 50-80 % of the Gaussians behind the camera) sorts a fraction of P, and the one digit all culled keys share is gone.
constexpr uint32_t kDropKey = kCulledKey;   // common.h: K1's depth key of a Gaussian without a tile
template <int kSortItems>
__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n_host, const uint32_t* __restrict__ n_dev, int drop,
                                                             int shift, int bits, uint32_t* __restrict__ hist, int nblocks) {
    constexpr int kSortTile = kRsThreads * kSortItems;
    __shared__ uint32_t s_h[kRsThreads / 64][kRsMaxBins];   // one histogram per wave: a quarter of the LDS-atomic collisions (-3 us per sort)
    const int tid = threadIdx.x, bins = 1 << bits, w = tid >> 6;
    const uint32_t mask = (uint32_t)bins - 1u;
    const uint32_t n = n_dev ? min(*n_dev, n_host) : n_host;
    const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
    if (base >= n) return;   // (the row scan does not read this block's column)
    for (int k = tid; k < (kRsThreads / 64) * kRsMaxBins; k += kRsThreads) (&s_h[0][0])[k] = 0;
    uint32_t key[kSortItems];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) { const uint32_t idx = base + (uint32_t)(i * kRsThreads + tid); key[i] = idx < n ? keys[idx] : kDropKey; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const uint32_t idx = base + (uint32_t)(i * kRsThreads + tid);
        if (idx < n && !(drop && key[i] == kDropKey)) atomicAdd(&s_h[w][(key[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (tid < bins) hist[(size_t)tid * nblocks + blockIdx.x] = (s_h[0][tid] + s_h[1][tid]) + (s_h[2][tid] + s_h[3][tid]);
}

// ---------------------------------------------------------------------------------------------
// ONE-SWEEP passes (round 6; 32-bit keys in four 8-bit passes, from kOneSweepFrom items up).  The classic pass above is three launches in a
// dependency chain -- histogram, row scan, scatter -- i.e. twelve per depth sort, each a few microseconds of work behind a launch gap.
// Here ONE kernel counts the digits of all four passes up front (the digit totals of a pass do not depend on the order the earlier passes
// leave the keys in), and every scatter block obtains "items of my digits in all EARLIER blocks" itself, by decoupled look-back over
// status words the blocks publish: block ids come from a ticket (so every predecessor is running or done), a status word carries tag,
// flag and value in 32 bits -- tag = the pass, so one table serves all four passes zeroed once; flag = partial (the block's own count)
// or inclusive (count + everything before it) -- and is written and polled with RELAXED agent-scope atomics: nothing else has to be
// ordered against it, and acquire / release at agent scope cost an L2 write-back / invalidate each on this chip of eight L2s
// (tools/ubench/lookback_ubench.hip: 9 us per pass of 730 blocks relaxed, 740 us with acquire / release).  Same positions as the classic
// pass, hence the same lists, bit for bit.  Six launches per sort instead of twelve.
// MEASURED (C3, 3 M keys, profiles/r06_one_sweep_sort.txt) and therefore OFF by default (SR_FLAG_ONE_SWEEP_SORT selects it): the depth sort
// takes 0.190 ms this way against 0.157 ms classic -- the look-back costs a scatter pass 13 us in situ (27 -> 40 us: all blocks finish
// counting at the same moment and walk back over each other's partial words), exactly what the histogram + row-scan kernels it replaces
// cost (7.9 + 4.9 us), and the up-front count of all four digits (23 us: four LDS atomics per key, the top byte takes four values) comes
// on top.  Kept: built, tested bit-exact on every ranking path, and the number to beat for whoever has a cheaper look-back.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kOneSweepFrom = 1u << 18;
constexpr int kSortOneSweepBit = 0x100;   // in the `rank_mode` argument of radix_sort_pairs
constexpr uint32_t kSwInclusive = 1u << 28, kSwValue = (1u << 28) - 1u;   // word = tag << 29 | inclusive << 28 | value
template <int kSortItems>
__global__ __launch_bounds__(kRsThreads) void rs_hist_all_kernel(const uint32_t* __restrict__ keys, uint32_t n, int drop, uint32_t* __restrict__ ghist) {
    constexpr int kSortTile = kRsThreads * kSortItems;
    __shared__ uint32_t s_h[kRsThreads / 64][4][kRsMaxBins];   // per wave and pass: 16 KB
    const int tid = threadIdx.x, w = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)kSortTile;
    for (int k = tid; k < (kRsThreads / 64) * 4 * kRsMaxBins; k += kRsThreads) (&s_h[0][0][0])[k] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const uint32_t idx = base + (uint32_t)(i * kRsThreads + tid);
        if (idx < n) {
            const uint32_t key = keys[idx];
            if (!(drop && key == kDropKey)) {
                atomicAdd(&s_h[w][0][key & 255u], 1u); atomicAdd(&s_h[w][1][(key >> 8) & 255u], 1u);
                atomicAdd(&s_h[w][2][(key >> 16) & 255u], 1u); atomicAdd(&s_h[w][3][key >> 24], 1u);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t c = (s_h[0][p][tid] + s_h[1][p][tid]) + (s_h[2][p][tid] + s_h[3][p][tid]);
        if (c) atomicAdd(&ghist[p * kRsMaxBins + tid], c);
    }
}

// Exclusive scan of every row (one block per digit), row totals out.
__global__ __launch_bounds__(kRsThreads) void rs_scan_rows_kernel(uint32_t* __restrict__ hist, int stride, const uint32_t* __restrict__ n_dev, uint32_t n_host,
                                                                  int items_per_block, uint32_t* __restrict__ row_total) {
    __shared__ uint32_t s_w[kRsThreads / 64];
    __shared__ uint32_t s_carry;
    const uint32_t n = n_dev ? min(*n_dev, n_host) : n_host;
    const int nblocks = (int)((n + (uint32_t)items_per_block - 1) / (uint32_t)items_per_block);   // (<= stride: the blocks that hold items)
    uint32_t* row = hist + (size_t)blockIdx.x * stride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int c = 0; c < nblocks; c += kRsThreads) {
        const int i = c + tid;
        const uint32_t x = i < nblocks ? row[i] : 0u;
        uint32_t incl = wave_inclusive_scan(x);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < w; ++k) wbase += s_w[k];
        const uint32_t carry = s_carry;
        if (i < nblocks) row[i] = carry + wbase + incl - x;
        __syncthreads();
        if (tid == kRsThreads - 1) s_carry = carry + wbase + incl;
        __syncthreads();
    }
    if (tid == 0) row_total[blockIdx.x] = s_carry;
}

// The 8-B payload of the depth sort (the tile rectangle, preprocess.hip: minx | miny << 16, width | height << 16) as ONE word that
// travels with the value through every pass: x fields of bx bits, y fields of by bits, 2 (bx + by) <= 32.
__device__ __forceinline__ uint32_t pack_rect(uint2 r, int bx, int by) {
    return (r.x & 0xFFFFu) | ((r.y & 0xFFFFu) << bx) | ((r.x >> 16) << (2 * bx)) | ((r.y >> 16) << (2 * bx + by));
}
__device__ __forceinline__ uint2 unpack_rect(uint32_t w, int bx, int by) {
    const uint32_t mx = (1u << bx) - 1u, my = (1u << by) - 1u;
    return make_uint2((w & mx) | (((w >> (2 * bx)) & my) << 16), ((w >> bx) & mx) | ((w >> (2 * bx + by)) << 16));
}

// kWide: the value is a pair (vals, ride) of words in two arrays; the first pass packs `ride` from aux_src[i] (read in order), the last
// unpacks it into aux_out -- instead of the last pass GATHERING aux_src[vals_out[i]]: at 3 M items that gather is 3 M random line fetches,
// 43 us of a 67 us pass (the other passes take 20), against ~3 us per pass for the extra word.  The ride is reordered through the same LDS
// buffer as the value, after it (the block keeps its 38 KB of LDS: four blocks per CU).
template <int kBits, bool kAtomicRank, int kSortItems, bool kWide, bool kOneSweep = false>   // digit width (compile time; 0 = run-time width <= 8), ranking (common.h take_run_slot), items per thread
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                uint32_t n_host, const uint32_t* __restrict__ n_dev, int drop, int shift, int bits_rt,
                                                                const uint32_t* __restrict__ hist,
                                                                const uint32_t* __restrict__ row_total, int nblocks,
                                                                const uint2* __restrict__ aux_src, uint2* __restrict__ aux_out,
                                                                const uint32_t* __restrict__ ride_in, uint32_t* __restrict__ ride_out, int bx, int by,
                                                                uint32_t* __restrict__ sw_status = nullptr, uint32_t* __restrict__ sw_ticket = nullptr, uint32_t sw_tag = 0) {
    // kOneSweep: `hist` is not read, `row_total` = the pass's 256 digit totals (rs_hist_all_kernel), sw_status [blocks][256] the look-back
    // table, sw_ticket this pass's block counter, sw_tag = (pass + 1) << 29
    constexpr int kSortTile = kRsThreads * kSortItems;
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_run[kRsThreads / 64][kRsMaxBins];    // items of digit b held by wave w; then, in place, the next block-local slot for (wave, digit)
    __shared__ uint32_t s_lstart[kRsMaxBins];                  // block-local start of digit b
    __shared__ uint32_t s_gbase[kRsMaxBins];                   // global position of the block's first item of digit b
    __shared__ uint32_t s_wsum[kRsThreads / 64];
    __shared__ uint32_t s_live;
    static_assert(kRsThreads / 64 == 4, "s_live sums four wave totals");
    __shared__ uint32_t s_key[kSortTile], s_val[kSortTile];        // the tile, stably reordered by digit
    const int bits = kBits ? kBits : bits_rt;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    const uint32_t n = n_dev ? min(*n_dev, n_host) : n_host;
    if (kOneSweep) {   // the block's id = the order in which it STARTED: every block with a smaller id is running or done
        if (tid == 0) s_bid = atomicAdd(sw_ticket, 1u);
        __syncthreads();
    }
    const uint32_t bid = kOneSweep ? s_bid : blockIdx.x;
    const uint32_t tile_base = bid * (uint32_t)kSortTile;
    if (tile_base >= n) return;   // (one-sweep: no later block looks back at a block without items)
    for (int b = tid; b < (kRsThreads / 64) * kRsMaxBins; b += kRsThreads) (&s_run[0][0])[b] = 0;
    __syncthreads();
    // wave w owns items [wbase, wbase + 512), 64 at a time in order -> stable
    const uint32_t wbase = tile_base + (uint32_t)w * (64 * kSortItems);
    uint32_t key[kSortItems], val[kSortItems], ride[kWide ? kSortItems : 1], slot[kWide ? kSortItems : 1];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const uint32_t idx = wbase + (uint32_t)(i * 64 + lane);
        key[i] = kDropKey; val[i] = 0;
        if constexpr (kWide) ride[i] = 0;
        if (idx < n) {
            key[i] = keys_in[idx];
            val[i] = vals_in ? vals_in[idx] : idx;   // first pass of an index sort: the value is the position itself
            if constexpr (kWide) ride[i] = ride_in ? ride_in[idx] : pack_rect(aux_src[idx], bx, by);
            if (!(drop && key[i] == kDropKey)) atomicAdd(&s_run[w][(key[i] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    // block-local exclusive scan over digits (thread t <-> digit t), then per-wave run starts
    {
        uint32_t tot = 0;
        if (tid < bins) {
#pragma unroll
            for (int k = 0; k < kRsThreads / 64; ++k) tot += s_run[k][tid];
        }
        uint32_t incl = wave_inclusive_scan(tot);
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        uint32_t off = incl - tot;
        for (int k = 0; k < w; ++k) off += s_wsum[k];
        if (tid == 0) s_live = (s_wsum[0] + s_wsum[1]) + (s_wsum[2] + s_wsum[3]);   // the tile's items that take part (all of them unless `drop`)
        // first output position of every digit = exclusive scan of the row totals (<= 256 values, L2-hot: cheaper here, in every
        // block, than as a kernel of its own between the row scan and this one)
        const uint32_t rt = tid < bins ? row_total[tid] : 0u;
        uint32_t rincl = wave_inclusive_scan(rt);
        __syncthreads();   // s_wsum is reused
        if (lane == 63) s_wsum[w] = rincl;
        __syncthreads();
        uint32_t bin_base = rincl - rt;
        for (int k = 0; k < w; ++k) bin_base += s_wsum[k];
        uint32_t before = 0;   // items of digit `tid` in all earlier blocks
        if (kOneSweep) {
            if (tid < bins) {
                uint32_t* mine = sw_status + (size_t)bid * kRsMaxBins + tid;
                __hip_atomic_store(mine, sw_tag | (bid == 0 ? kSwInclusive : 0u) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bid > 0) {
                    for (int p = (int)bid - 1; p >= 0; --p) {
                        uint32_t v;
                        while ((((v = __hip_atomic_load(sw_status + (size_t)p * kRsMaxBins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 29) << 29) != sw_tag)
                            __builtin_amdgcn_s_sleep(1);
                        before += v & kSwValue;
                        if (v & kSwInclusive) break;
                    }
                    __hip_atomic_store(mine, sw_tag | kSwInclusive | (before + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else if (tid < bins) {
            before = hist[(size_t)tid * nblocks + blockIdx.x];
        }
        if (tid < bins) {
            s_lstart[tid] = off;
            s_gbase[tid] = bin_base + before;
            uint32_t run = off;
#pragma unroll
            for (int k = 0; k < kRsThreads / 64; ++k) { const uint32_t c = s_run[k][tid]; s_run[k][tid] = run; run += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const uint32_t idx = wbase + (uint32_t)(i * 64 + lane);
        const bool live = idx < n && !(drop && key[i] == kDropKey);
        const uint32_t d = (key[i] >> shift) & mask;
        // rank among the wave's items of the same digit, in item order, and the advance of the (wave, digit) run (common.h)
        const uint32_t pos = take_run_slot<kAtomicRank>(s_run[w], d, live, bits);
        if (live) { s_key[pos] = key[i]; s_val[pos] = val[i]; }
        if constexpr (kWide) slot[i] = pos;
    }
    __syncthreads();
    // coalesced write-out: consecutive local slots of one digit are consecutive in global memory
    const uint32_t count = s_live;   // (= min(kSortTile, n - tile_base) unless items were dropped)
    uint32_t dest[kWide ? kSortItems : 1];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const uint32_t li = (uint32_t)(i * kRsThreads + tid);
        if (li < count) {
            const uint32_t k = s_key[li];
            const uint32_t d = (k >> shift) & mask;
            const uint32_t g = s_gbase[d] + (li - s_lstart[d]);
            const uint32_t v = s_val[li];
            if constexpr (kWide) dest[i] = g;
            keys_out[g] = k;
            vals_out[g] = v;
            if (!kWide && aux_out) aux_out[g] = aux_src[v];   // last pass: payload gathered in sorted order (aux_out[i] = aux_src[vals_out[i]])
        }
    }
    if constexpr (kWide) {   // the ride, through s_val once the values are out
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSortItems; ++i)
            if (wbase + (uint32_t)(i * 64 + lane) < n && !(drop && key[i] == kDropKey)) s_val[slot[i]] = ride[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kSortItems; ++i) {
            const uint32_t li = (uint32_t)(i * kRsThreads + tid);
            if (li < count) {
                if (ride_out) ride_out[dest[i]] = s_val[li];
                else aux_out[dest[i]] = unpack_rect(s_val[li], bx, by);   // last pass
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Emission offsets (K2): the duplicates of Gaussian i own the gradient-record slots first[i] .. first[i] + tiles_touched[i] - 1,
// first = exclusive scan of tiles_touched in ID order (any order would do; this one needs no gather and no scattered store).  Two small
// kernels: an exclusive scan inside every block of 2048 Gaussians + the block totals; then an exclusive scan of the totals.  Consumers
// add the block base themselves (common.h first_index()); totals[nblocks] = D, the number of duplicates, is what the host reads back.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRsThreads) void scan_blocks_kernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ block_total) {
    __shared__ uint32_t s_w[kRsThreads / 64], s_nz[kRsThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)kRsTile + (uint32_t)tid * kRsItems;   // 8 consecutive items per thread
    uint32_t c[kRsItems];
    if (base + kRsItems <= n) {   // two 16-B loads
        const uint4 c0 = reinterpret_cast<const uint4*>(counts + base)[0], c1 = reinterpret_cast<const uint4*>(counts + base)[1];
        c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
    } else {
#pragma unroll
        for (int i = 0; i < kRsItems; ++i) c[i] = base + i < n ? counts[base + i] : 0u;
    }
    uint32_t v[kRsItems], sum = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < kRsItems; ++i) {
        v[i] = sum;   // exclusive within the thread
        sum += c[i];
        nz += c[i] != 0u;
    }
    uint32_t incl = wave_inclusive_scan(sum);
    const uint32_t nz_incl = wave_inclusive_scan(nz);
    if (lane == 63) { s_w[w] = incl; s_nz[w] = nz_incl; }
    __syncthreads();
    uint32_t off = incl - sum;
    for (int k = 0; k < w; ++k) off += s_w[k];
    if (base + kRsItems <= n) {
        reinterpret_cast<uint4*>(out + base)[0] = make_uint4(off + v[0], off + v[1], off + v[2], off + v[3]);
        reinterpret_cast<uint4*>(out + base)[1] = make_uint4(off + v[4], off + v[5], off + v[6], off + v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < kRsItems; ++i) if (base + i < n) out[base + i] = off + v[i];
    }
    if (tid == kRsThreads - 1) {
        block_total[blockIdx.x] = off + sum;
        // side product: the Gaussians of this block with at least one tile -- their total, the frame's VISIBLE count, travels to the host
        // with D and picks the forward blend kernel (api.hip: duplicates per visible Gaussian)
        uint32_t n_vis = 0;
        for (int k = 0; k < kRsThreads / 64; ++k) n_vis += s_nz[k];
        block_total[gridDim.x + 2 + blockIdx.x] = n_vis;
    }
}

__global__ __launch_bounds__(kRsThreads) void scan_totals_kernel(uint32_t* __restrict__ block_total, int nblocks, uint32_t* __restrict__ total_host) {
    // exclusive scan in place, one block, sequential over chunks of 256; block_total[nblocks] = grand total (also stored straight into
    // the caller's pinned host words when given: no copy kernel between this one and the host's wake-up); block_total[nblocks + 1] /
    // total_host[1] = the sum of the blocks' non-zero counts (scan_blocks_kernel), i.e. the number of items with a non-zero count
    __shared__ uint32_t s_w[kRsThreads / 64];
    __shared__ uint32_t s_carry, s_vis;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { s_carry = 0; s_vis = 0; }
    __syncthreads();
    uint32_t vis_mine = 0;   // (this thread's share of the blocks' non-zero counts, fetched next to the totals)
    for (int c = 0; c < nblocks; c += kRsThreads) {
        const int i = c + tid;
        const uint32_t x = i < nblocks ? block_total[i] : 0u;
        vis_mine += i < nblocks ? block_total[nblocks + 2 + i] : 0u;
        uint32_t incl = wave_inclusive_scan(x);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t wb = 0;
        for (int k = 0; k < w; ++k) wb += s_w[k];
        const uint32_t carry = s_carry;
        if (i < nblocks) block_total[i] = carry + wb + incl - x;
        __syncthreads();
        if (tid == kRsThreads - 1) s_carry = carry + wb + incl;
        __syncthreads();
    }
    {
        const uint32_t tot = wave_inclusive_scan(vis_mine);
        if (lane == 63 && tot) atomicAdd(&s_vis, tot);
        __syncthreads();
    }
    if (tid == 0) { block_total[nblocks] = s_carry; block_total[nblocks + 1] = s_vis; if (total_host) { total_host[0] = s_carry; total_host[1] = s_vis; } }
}

// Run-time guard of the rank phase (api.hip rank_mode, once per device): 64 rounds of 64 items per wave with alphabets from 1 to 1000
// digits, ranked three ways on running counters -- by LDS atomic returns, by the match-any ballots, and by plain LDS loads / stores
// (every lane counts the lower lanes holding its digit; the reference).  *result: bit 0 = the atomic returns are the stable ranks,
// bit 1 = the ballot ranks are.
__global__ __launch_bounds__(kRsThreads) void rank_selfcheck_kernel(uint32_t* __restrict__ result) {
    __shared__ uint32_t s_a[kRsThreads / 64][1024], s_b[kRsThreads / 64][1024], s_c[kRsThreads / 64][1024];
    __shared__ uint32_t s_dig[kRsThreads / 64][64];
    __shared__ uint32_t s_ok;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < (kRsThreads / 64) * 1024; i += kRsThreads) { (&s_a[0][0])[i] = 0; (&s_b[0][0])[i] = 0; (&s_c[0][0])[i] = 0; }
    if (tid == 0) s_ok = 3u;
    __syncthreads();
    uint32_t ok = 3u;
    for (int round = 0; round < 64; ++round) {
        const uint32_t alphabets[8] = {1u, 2u, 3u, 16u, 64u, 256u, 700u, 1000u};
        const uint32_t bins = alphabets[round & 7];
        uint32_t h = (uint32_t)(round * 64 + lane) * 2654435761u + (uint32_t)w * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t d = (round & 8) ? (h % bins) : ((h >> 3) % ((bins + 3u) / 4u));   // (also heavily repeated digits)
        const bool live = (round & 16) ? lane < 37 : true;                                // (and partially filled rows)
        const uint32_t pa = take_run_slot<true>(s_a[w], d, live, 10);
        const uint32_t pb = take_run_slot<false>(s_b[w], d, live, 10);
        s_dig[w][lane] = live ? d : 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        uint32_t lower = 0, higher = 0;
        for (int k = 0; k < 64; ++k) { const uint32_t dk = s_dig[w][k]; lower += (k < lane && dk == d); higher += (k > lane && dk == d); }
        const uint32_t pc = s_c[w][d] + lower;
        __builtin_amdgcn_wave_barrier();
        if (live && higher == 0) s_c[w][d] = pc + 1u;   // the last lane of every digit group advances the reference counter
        __builtin_amdgcn_wave_barrier();
        if (live && pa != pc) ok &= ~1u;
        if (live && pb != pc) ok &= ~2u;
    }
    atomicAnd(&s_ok, ok);
    __syncthreads();
    if (tid == 0) *result = s_ok | 0x100u;   // bit 8: the kernel ran
}
hipError_t launch_rank_selfcheck(uint32_t* result, hipStream_t s) {
    hipLaunchKernelGGL(rank_selfcheck_kernel, dim3(1), dim3(kRsThreads), 0, s, result);
    return hipGetLastError();
}

// Test hook (sr_debug_lds_atomic_ranks): the lane order of LDS atomic returns, the property the rank phase above relies on.
__global__ __launch_bounds__(kRsThreads) void lds_atomic_ranks_kernel(const uint32_t* __restrict__ digits, uint32_t* __restrict__ ranks, uint32_t n, int bins) {
    __shared__ uint32_t s_cnt[kRsThreads / 64][1024];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < (kRsThreads / 64) * 1024; i += kRsThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    for (int it = 0; it < kRsItems; ++it) {
        const uint32_t idx = (blockIdx.x * (uint32_t)kRsItems + it) * kRsThreads + tid;
        if (idx < n) ranks[idx] = atomicAdd(&s_cnt[w][digits[idx] % (uint32_t)bins], 1u);
    }
}
hipError_t lds_atomic_ranks(const uint32_t* digits, uint32_t* ranks, uint32_t n, int bins, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lds_atomic_ranks_kernel, dim3((n + kRsTile - 1) / kRsTile), dim3(kRsThreads), 0, s, digits, ranks, n, bins);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRsThreads) void rs_zero_words_kernel(uint32_t* __restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * kRsThreads + threadIdx.x; i < n; i += gridDim.x * kRsThreads) p[i] = 0u;
}
static inline int rs_blocks(uint32_t n, int items = kRsItems) { return (int)((n + (uint32_t)(kRsThreads * items) - 1) / (uint32_t)(kRsThreads * items)); }
static inline int scan_blocks(uint32_t n) { return (int)((n + kRsTile - 1) / kRsTile); }

// scratch: ping-pong (keys, vals) + histogram table + row totals
size_t radix_sort_temp_bytes(uint32_t n) {
    const size_t nb = (size_t)rs_blocks(n > 0 ? n : 1);
    // (keys, vals, ride) | histogram table = the one-sweep status table | row totals, the four passes' digit totals, four tickets
    return align_up((size_t)(n > 0 ? n : 1) * 4, 256) * 3 + align_up(nb * kRsMaxBins * 4, 256) + align_up(kRsMaxBins * 4 * 5 + 64, 256);
}

// Sorts by key bits [0, total_bits) in passes of <= 8 bits (as even as possible); stable; result in keys_out/vals_out.
// vals_in == nullptr means "value = index".  keys_in / vals_in are not modified.  If aux_out != nullptr the sort also delivers
// aux_out[i] = aux_src[vals_out[i]] (an 8-B payload in sorted order): packed into a word that rides along with the value when it is
// a tile rectangle whose fields fit (value = index, fields < 2^rect_bx / 2^rect_by, 2 (bx + by) <= 32), gathered by the last pass otherwise.
// `n_live` != nullptr (device word): a COMPACTING sort -- items whose key is 0xFFFFFFFF are dropped by the first pass, *n_live must be the
// number of the others, and only keys_out / vals_out / aux_out[0 .. *n_live) are written.
hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint2* aux_src, uint2* aux_out, int rank_mode,
                            int rect_bx, int rect_by, const uint32_t* n_live) {
    if (n == 0) return hipSuccess;
    const bool sweep_wanted = (rank_mode & kSortOneSweepBit) != 0;   // (rides in the ranking argument: SR_FLAG_ONE_SWEEP_SORT)
    rank_mode &= ~kSortOneSweepBit;
    if (temp_bytes < radix_sort_temp_bytes(n) || (rank_mode != kRankAtomic && rank_mode != kRankBallot)) return hipErrorInvalidValue;
    int passes = (total_bits + 7) / 8;
    if (passes < 1) passes = 1;
    if (passes & 1) ++passes;   // even pass count: the ping-pong ends in keys_out/vals_out (an extra pass on zero bits is a stable copy)
    const int nb8 = rs_blocks(n);
    char* t = static_cast<char*>(temp);
    uint32_t* tk = reinterpret_cast<uint32_t*>(t); t += align_up((size_t)n * 4, 256);
    uint32_t* tv = reinterpret_cast<uint32_t*>(t); t += align_up((size_t)n * 4, 256);
    uint32_t* tr = reinterpret_cast<uint32_t*>(t); t += align_up((size_t)n * 4, 256);
    // the ride's other buffer is the first half of aux_out itself: written by odd passes, read by the next one; the last pass reads `tr`
    const bool wide = aux_out && !vals_in && rect_bx > 0 && rect_by > 0 && 2 * (rect_bx + rect_by) <= 32 && passes >= 2 && n >= kSortRideFrom;
    uint32_t* hist = reinterpret_cast<uint32_t*>(t); t += align_up((size_t)nb8 * kRsMaxBins * 4, 256);
    uint32_t* row_total = reinterpret_cast<uint32_t*>(t);
    uint32_t* ghist = row_total + kRsMaxBins;          // [4][256] digit totals of the four passes (one-sweep)
    uint32_t* tickets = ghist + 4 * kRsMaxBins;        // [4] block counters, one per pass
    const uint32_t* ki = keys_in; const uint32_t* vi = vals_in;
    int shift = 0, left = total_bits;
    const bool sweep = sweep_wanted && total_bits == 32 && passes == 4 && n >= kOneSweepFrom && n <= kSwValue;
    if (sweep) {
        // status table (in the histogram table's place: the same [blocks][256] words), row totals, digit totals, tickets: zeroed once per sort
        const size_t zero_bytes = (size_t)(reinterpret_cast<char*>(tickets + 16) - reinterpret_cast<char*>(hist));
        hipLaunchKernelGGL(rs_zero_words_kernel, dim3(256), dim3(kRsThreads), 0, s, hist, (uint32_t)(zero_bytes / 4));
        const bool bigh = n >= kSortBigFrom;
        if (bigh) hipLaunchKernelGGL(rs_hist_all_kernel<kSortItemsBig>, dim3(rs_blocks(n, kSortItemsBig)), dim3(kRsThreads), 0, s, keys_in, n, n_live ? 1 : 0, ghist);
        else hipLaunchKernelGGL(rs_hist_all_kernel<kRsItems>, dim3(nb8), dim3(kRsThreads), 0, s, keys_in, n, n_live ? 1 : 0, ghist);
    }
    for (int p = 0; p < passes; ++p) {
        const int remaining_passes = passes - p;
        int bits = left > 0 ? (left + remaining_passes - 1) / remaining_passes : 1;
        if (bits > 8) bits = 8;
        if (bits < 1) bits = 1;
        const bool last = p == passes - 1;
        uint32_t* ko = (p & 1) ? keys_out : tk;
        uint32_t* vo = (p & 1) ? vals_out : tv;
        const uint32_t* ri = p == 0 ? nullptr : ((p & 1) ? tr : reinterpret_cast<const uint32_t*>(aux_out));
        uint32_t* ro = last ? nullptr : ((p & 1) ? reinterpret_cast<uint32_t*>(aux_out) : tr);
        const bool big = bits == 8 && n >= kSortBigFrom;
        const int nb = big ? rs_blocks(n, kSortItemsBig) : nb8;
        const uint32_t* nd = (n_live && p > 0) ? n_live : nullptr;   // the first pass reads everything and drops; the later ones see the survivors
        const int drop = (n_live && p == 0) ? 1 : 0;
        if (!sweep) {
            if (big) hipLaunchKernelGGL(rs_hist_kernel<kSortItemsBig>, dim3(nb), dim3(kRsThreads), 0, s, ki, n, nd, drop, shift, bits, hist, nb);
            else hipLaunchKernelGGL(rs_hist_kernel<kRsItems>, dim3(nb), dim3(kRsThreads), 0, s, ki, n, nd, drop, shift, bits, hist, nb);
            hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(1 << bits), dim3(kRsThreads), 0, s, hist, nb, nd, n, kRsThreads * (big ? kSortItemsBig : kRsItems), row_total);
        }
#define SR_SCATTER_W(B, A, I, Wd) do { if (sweep && B == 8) hipLaunchKernelGGL((rs_scatter_kernel<B == 8 ? 8 : 0, A, I, Wd, B == 8>), dim3(nb), dim3(kRsThreads), 0, s, ki, vi, ko, vo, n, nd, drop, shift, bits, \
                                                 (const uint32_t*)nullptr, (const uint32_t*)(ghist + p * kRsMaxBins), nb, (Wd ? p == 0 : last) ? aux_src : nullptr, last ? aux_out : nullptr, ri, ro, rect_bx, rect_by, \
                                                 hist, tickets + p, (uint32_t)(p + 1) << 29);                                                                    \
                                   else hipLaunchKernelGGL((rs_scatter_kernel<B, A, I, Wd>), dim3(nb), dim3(kRsThreads), 0, s, ki, vi, ko, vo, n, nd, drop, shift, bits, (const uint32_t*)hist, \
                                                 (const uint32_t*)row_total, nb, (Wd ? p == 0 : last) ? aux_src : nullptr, last ? aux_out : nullptr, ri, ro, rect_bx, rect_by, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u); } while (0)
#define SR_SCATTER_R(B, A, I) do { if (wide) SR_SCATTER_W(B, A, I, true); else SR_SCATTER_W(B, A, I, false); } while (0)
#define SR_SCATTER(B, I) do { if (rank_mode == kRankAtomic) SR_SCATTER_R(B, true, I); else SR_SCATTER_R(B, false, I); } while (0)
        switch (bits) {
            case 8: if (big) SR_SCATTER(8, kSortItemsBig); else SR_SCATTER(8, kRsItems); break;
            case 7: SR_SCATTER(7, kRsItems); break;
            case 6: SR_SCATTER(6, kRsItems); break;
            default: SR_SCATTER(0, kRsItems); break;
        }
#undef SR_SCATTER
#undef SR_SCATTER_R
#undef SR_SCATTER_W
        ki = ko; vi = vo;
        shift += bits; left -= bits;
    }
    return hipGetLastError();
}

size_t tile_count_scan_temp_bytes(uint32_t n) { return align_up((2 * (size_t)scan_blocks(n > 0 ? n : 1) + 2) * 4, 256); }   // bases, D, visible, per-block visible

// out[i] = exclusive scan of counts INSIDE block i / 2048; block_base[b] = exclusive scan of the block totals,
// block_base[nblocks] = total.  (block_base lives in `temp`.)
hipError_t tile_count_scan(const uint32_t* counts, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes, uint32_t* total_host, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (temp_bytes < tile_count_scan_temp_bytes(n)) return hipErrorInvalidValue;
    const int nb = scan_blocks(n);
    uint32_t* totals = static_cast<uint32_t*>(temp);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(nb), dim3(kRsThreads), 0, s, counts, n, out, totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kRsThreads), 0, s, totals, nb, total_host);
    return hipGetLastError();
}

}  // namespace sr
