// binning.hip -- K2..K5: depth ordering of Gaussians, tile-count scan, expanding tile partition, tile ranges.
//
// The reference algorithm emits one (tile<<32 | depth_bits, gaussian)
// pair per touched tile and LSD-radix-sorts all D 64-bit keys.  The result is the list ordered by
// (tile, depth bits, gaussian id).  This build produces the *same list, bit for bit*, without ever materialising the
// (tile, id) pairs:
//   1. sort the P Gaussians once by their 32-bit depth key (stable => ties keep ascending id);
//   2. scan the tile counts (in id order): the duplicates of one Gaussian own the contiguous "emission" indices
//      first[gid] + (ty - miny) * w + (tx - minx), which is where K7 stores the per-duplicate gradient records so that
//      K8 can sum them as one contiguous span (and the total D is the one word the host reads back);
//   3. partition by tile column while expanding rectangles along x, then by tile row while expanding along y
//      (the "expanding partition" below); pass Y also counts the entries per tile, so the tile ranges are one
//      exclusive scan of 8 160 counters.
// Stability of every step makes (tile, depth, id) the final order.  Integer work only.
#include "common.h"

namespace sr {

// radix_sort.hip
size_t radix_sort_temp_bytes(uint32_t n);
hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint2* aux_src, uint2* aux_out, int rank_mode,
                            int rect_bx = 0, int rect_by = 0, const uint32_t* n_live = nullptr);
size_t tile_count_scan_temp_bytes(uint32_t n);
hipError_t tile_count_scan(const uint32_t* counts, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes, uint32_t* total_host, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// K3 + K4: the "expanding partition".  The final list is ordered by (tile row, tile column, depth rank), so an LSD partition runs
// over the tile column first and the tile row second -- and a Gaussian's rectangle only has to be expanded along the axis a pass
// partitions by:
//   pass X: the P Gaussians in depth order, each expanded into its w tile columns, stably partitioned by column
//           -> "column items" (gaussian id, first row, rows, column), sum(w) of them (~ D / 2.3);
//   pass Y: the column items, each expanded into its h rows, stably partitioned by row -> the D entries of point_list.
// No duplicate is ever written in unsorted form: the only D-sized traffic is the final 4-B store per entry.  (Emitting all D (tile, id) pairs and running two radix passes over them moved 3.5x the bytes.)
// One pass = histogram -> row scan -> scatter, like the radix sort's; what differs is that the "keys" of a block are generated, not
// loaded: a block owns 1024 / 2048 input items; a block-local prefix sum of their lengths gives every item its first slot, the items
// mark those slots in LDS, and a ballot of the marks + a popcount map a row of 64 slots back to (item, offset); the slots are then
// ranked (one LDS atomic each, radix_sort.hip), reordered in LDS and written out in batches of 2048.  The histogram needs no
// expansion at all: +1 at the first digit of an item, -1 behind its last, prefix sum (LDS).
// Pass Y's histogram kernel also produces the number of entries per tile (the input is ordered by column, so a block sees one or two
// columns: 4 x rows difference counters in LDS), whose exclusive scan is the range table (K5).
// ---------------------------------------------------------------------------------------------
constexpr int kXpThreads = 256, kXpWaves = kXpThreads / 64;
constexpr int kXpInputsX = 1024, kXpInputsY = 2048;   // input items per block (pass X has only P / 1024 ~ 11 blocks per CU as it is)
template <int AXIS> constexpr int xp_inputs() { return AXIS == 0 ? kXpInputsX : kXpInputsY; }
constexpr int kXpBatch = 2048;     // expanded slots ranked + reordered in LDS at a time
constexpr int kXpMaxBins = 1024;   // tiles per image axis
// column item: x = gaussian id, y = first row | rows << 10 | column << 21
__device__ inline uint32_t pack_column(int miny, int h, int tx) { return (uint32_t)miny | ((uint32_t)h << 10) | ((uint32_t)tx << 21); }

// exclusive prefix of x over the block's threads (+ the block total); s_w: kXpWaves words of scratch
__device__ inline uint32_t block_exclusive_scan(uint32_t x, uint32_t* s_w, uint32_t& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_inclusive_scan(x);
    __syncthreads();   // s_w may still be in use from the previous scan
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kXpWaves; ++k) { const uint32_t c = s_w[k]; if (k < w) off += c; tot += c; }
    total = tot;
    return off + incl - x;
}

template <int AXIS> __device__ inline void expansion_of(uint2 item, int& start, int& len) {
    if (AXIS == 0) { start = (int)(item.x & 0xFFFFu); len = (int)(item.y & 0xFFFFu); }            // rect: minx | miny << 16, w | h << 16
    else { start = (int)(item.y & 1023u); len = (int)((item.y >> 10) & 2047u); }                  // column item
}

// hist[d * stride + block] = expanded slots of digit d in this block.  n_dev != NULL: the item count lives on the device (pass Y).
template <int AXIS>
__global__ __launch_bounds__(kXpThreads) void expand_hist_kernel(const uint2* __restrict__ in, uint32_t n_host, const uint32_t* __restrict__ n_dev,
                                                                 int bins, uint32_t* __restrict__ hist, int stride, int tiles_x,
                                                                 uint32_t* __restrict__ tile_counts, int n_tiles) {
    __shared__ uint32_t s_diff[kXpMaxBins + 1];
    __shared__ uint32_t s_full[AXIS ? 4 : 1][AXIS ? kXpMaxBins + 1 : 1];
    __shared__ uint32_t s_w[kXpWaves];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (AXIS == 0)   // side job: clear the per-tile counters pass Y accumulates into
        for (int k = blockIdx.x * kXpThreads + tid; k < n_tiles; k += gridDim.x * kXpThreads) tile_counts[k] = 0;
    const uint32_t n = n_dev ? *n_dev : n_host;
    constexpr int kInputs = xp_inputs<AXIS>();
    const uint32_t base = blockIdx.x * (uint32_t)kInputs;
    if (base >= n) return;
    for (int k = tid; k <= bins; k += kXpThreads) s_diff[k] = 0;
    int tx0 = 0;
    if (AXIS == 1) {
        for (int k = tid; k < 4 * (kXpMaxBins + 1); k += kXpThreads) (&s_full[0][0])[k] = 0;
        tx0 = (int)(in[base].y >> 21);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kInputs / kXpThreads; ++i) {
        const uint32_t idx = base + (uint32_t)(i * kXpThreads + tid);
        if (idx < n) {
            const uint2 it = in[idx];
            int start, len; expansion_of<AXIS>(it, start, len);
            if (len) {
                atomicAdd(&s_diff[start], 1u); atomicSub(&s_diff[start + len], 1u);
                if (AXIS == 1) {
                    const int tx = (int)(it.y >> 21);
                    const uint32_t rel = (uint32_t)(tx - tx0);
                    if (rel < 4u) { atomicAdd(&s_full[rel][start], 1u); atomicSub(&s_full[rel][start + len], 1u); }
                    else for (int k = 0; k < len; ++k) atomicAdd(&tile_counts[(start + k) * tiles_x + tx], 1u);   // tiny inputs only
                }
            }
        }
    }
    __syncthreads();
    {   // prefix sum over the digits, 4 consecutive digits per thread
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = 4 * tid + j; sum += b < bins ? s_diff[b] : 0u; v[j] = sum; }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(sum, s_w, total);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = 4 * tid + j; if (b < bins) hist[(size_t)b * stride + blockIdx.x] = excl + v[j]; }
    }
    if (AXIS == 1 && tx0 + w < tiles_x) {   // wave w: the rows of column tx0 + w
        uint32_t carry = 0;
        for (int c = 0; c < bins; c += 64) {
            const int row = c + lane;
            uint32_t incl = wave_inclusive_scan(row < bins ? s_full[w][row] : 0u);
            const uint32_t cnt = carry + incl;
            if (row < bins && cnt) atomicAdd(&tile_counts[row * tiles_x + tx0 + w], cnt);
            carry += (uint32_t)__shfl((int)incl, 63);
        }
    }
}

// Exclusive scan of every histogram row (one block per digit) + the row totals.
__global__ __launch_bounds__(kXpThreads) void expand_scan_rows_kernel(uint32_t* __restrict__ hist, int stride, uint32_t n_host,
                                                                      const uint32_t* __restrict__ n_dev, int inputs_per_block,
                                                                      uint32_t* __restrict__ row_total) {
    __shared__ uint32_t s_w[kXpWaves];
    const uint32_t n = n_dev ? *n_dev : n_host;
    const int nblocks = (int)((n + (uint32_t)inputs_per_block - 1) / (uint32_t)inputs_per_block);
    uint32_t* row = hist + (size_t)blockIdx.x * stride;
    uint32_t carry = 0;
    for (int c = 0; c < nblocks; c += kXpThreads) {
        const int i = c + (int)threadIdx.x;
        const uint32_t x = i < nblocks ? row[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(x, s_w, total);
        if (i < nblocks) row[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

template <int AXIS, int kBits, bool kAtomicRank>
__global__ __launch_bounds__(kXpThreads) void expand_scatter_kernel(
    const uint2* __restrict__ in, uint32_t n_host, const uint32_t* __restrict__ n_dev, int bins, const uint32_t* __restrict__ hist, int stride,
    const uint32_t* __restrict__ row_total,
    // pass X: ids in depth order, outputs
    const uint32_t* __restrict__ sorted_gid, uint2* __restrict__ columns_out, uint32_t* __restrict__ n_columns_out,
    // pass Y
    uint32_t* __restrict__ point_list) {
    constexpr int kBins = 1 << kBits;
    constexpr int kPer = kBins / kXpThreads > 0 ? kBins / kXpThreads : 1;   // digits per thread in the block-wide digit scans
    constexpr int kInputs = xp_inputs<AXIS>(), kXpPerThread = kInputs / kXpThreads;
    __shared__ __attribute__((aligned(16))) uint8_t s_start[kXpBatch];                 // 1 where a slot of the current batch is the first slot of an item
    __shared__ uint32_t s_prefix[kInputs + 1];            // s_prefix[j] = expanded slots of the block's items before item j
    __shared__ uint32_t s_count[kXpWaves][kBins];         // slots of digit d held by wave w (this batch); then, in place, the next
                                                          // batch-local position for (wave, digit)
    __shared__ uint32_t s_lstart[kBins];                  // batch-local start of digit d
    __shared__ uint32_t s_goff[kBins];                    // global position of the block's next slot of digit d
    __shared__ uint32_t s_w[kXpWaves];
    __shared__ uint32_t s_id[kXpBatch];                   // the batch, stably reordered by digit: gaussian id,
    __shared__ uint32_t s_rows[AXIS == 0 ? kXpBatch : 1]; //   first row | rows << 10 (pass X),
    __shared__ uint16_t s_dig[kXpBatch];                  //   digit
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = n_dev ? *n_dev : n_host;
    const uint32_t base = blockIdx.x * (uint32_t)kInputs;
    if (base >= n) return;
    const int n_items = (int)min((uint32_t)kInputs, n - base);
    uint32_t E;   // expanded slots of this block
    uint32_t it_first[kXpPerThread];   // first slot of this thread's items (0xFFFFFFFF: an empty item)
    {
        uint32_t incl[kXpPerThread], sum = 0;
#pragma unroll
        for (int i = 0; i < kXpPerThread; ++i) {
            const int j = tid * kXpPerThread + i;
            int start = 0, len = 0;
            if (j < n_items) {
                const uint2 it = in[base + j];
                expansion_of<AXIS>(it, start, len);
            }
            it_first[i] = len ? sum : 0xFFFFFFFFu;
            sum += (uint32_t)len;
            incl[i] = sum;
        }
        for (int k = tid; k < kXpBatch / 4; k += kXpThreads) reinterpret_cast<uint32_t*>(s_start)[k] = 0;
        const uint32_t excl = block_exclusive_scan(sum, s_w, E);
#pragma unroll
        for (int i = 0; i < kXpPerThread; ++i) {
            s_prefix[tid * kXpPerThread + i + 1] = excl + incl[i];
            if (it_first[i] != 0xFFFFFFFFu) it_first[i] += excl;
        }
        if (tid == 0) s_prefix[0] = 0;
    }
    {   // first output position of every digit (exclusive scan of the row totals) + this block's offset inside the digit
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) { const int d = tid * kPer + j; const uint32_t c = (d < bins) ? row_total[d] : 0u; v[j] = sum; sum += c; }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(sum, s_w, total);
#pragma unroll
        for (int j = 0; j < kPer; ++j) { const int d = tid * kPer + j; if (d < kBins) s_goff[d] = d < bins ? excl + v[j] + hist[(size_t)d * stride + blockIdx.x] : 0u; }
        if (AXIS == 0 && blockIdx.x == 0 && tid == 0) *n_columns_out = total;   // number of column items = pass Y's input size
    }
    for (uint32_t q0 = 0; q0 < E; q0 += kXpBatch) {
        for (int k = tid; k < kXpWaves * kBins; k += kXpThreads) (&s_count[0][0])[k] = 0;
        // every non-empty item marks its first slot (the marks of the previous batch were cleared behind its last reader)
#pragma unroll
        for (int i = 0; i < kXpPerThread; ++i)
            if (it_first[i] - q0 < (uint32_t)kXpBatch) s_start[it_first[i] - q0] = 1;   // (an empty item's 0xFFFFFFFF never lands in a batch)
        __syncthreads();   // (also orders s_prefix / s_goff writes and the previous batch's write-out)
        // slot -> item.  There are no empty items (the depth sort drops the culled Gaussians; a column item has >= 1 row) -- and trailing ones would do no harm --, so
        // the item of a slot is the number of marks up to it, minus one: one binary search per wave and batch for the wave's first
        // slot, then a ballot of the marks and a popcount per row of 64 slots.
        const uint32_t s0 = q0 + (uint32_t)(w * (kXpBatch / kXpWaves));
        uint32_t marks_before = 0;   // marks in front of the current row (block-wide count)
        if (s0 < E) {
            int lo = 0;   // largest j with s_prefix[j] <= s0
#pragma unroll
            for (int step = kInputs / 2; step > 0; step >>= 1)
                if (lo + step < n_items && s_prefix[lo + step] <= s0) lo += step;
            marks_before = (uint32_t)lo + (s_prefix[lo] == s0 ? 0u : 1u);
        }
        uint32_t dig[kXpBatch / kXpThreads], id[kXpBatch / kXpThreads], rows[AXIS == 0 ? kXpBatch / kXpThreads : 1];
#pragma unroll
        for (int i = 0; i < kXpBatch / kXpThreads; ++i) {
            const uint32_t slot = s0 + (uint32_t)(i * 64 + lane);
            dig[i] = 0; id[i] = 0;
            if (AXIS == 0) rows[i] = 0;
            const bool mark = slot < E && s_start[slot - q0] != 0;
            const unsigned long long marks = ballot64(mark);
            const uint32_t lo = marks_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u)) +
                                (mark ? 1u : 0u) - 1u;
            marks_before += (uint32_t)__popcll(marks);
            if (slot < E) {
                const uint2 it = in[base + lo];
                int start, len; expansion_of<AXIS>(it, start, len);
                dig[i] = (uint32_t)start + (slot - s_prefix[lo]);
                if (AXIS == 0) { id[i] = sorted_gid[base + lo]; rows[i] = (it.x >> 16) | ((it.y >> 16) << 10); }
                else id[i] = it.x;
                atomicAdd(&s_count[w][dig[i]], 1u);
            }
        }
        __syncthreads();
        for (int k = tid; k < kXpBatch / 4; k += kXpThreads) reinterpret_cast<uint32_t*>(s_start)[k] = 0;   // for the next batch
        uint32_t tot[kPer];
        {   // batch-local exclusive scan over the digits, then the per-wave run starts
            uint32_t sum = 0;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int d = tid * kPer + j;
                tot[j] = 0;
                if (d < kBins) {
#pragma unroll
                    for (int k = 0; k < kXpWaves; ++k) tot[j] += s_count[k][d];
                }
                sum += tot[j];
            }
            uint32_t total;
            uint32_t run = block_exclusive_scan(sum, s_w, total);
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int d = tid * kPer + j;
                if (d < kBins) {
                    s_lstart[d] = run;
#pragma unroll
                    for (int k = 0; k < kXpWaves; ++k) { const uint32_t c = s_count[k][d]; s_count[k][d] = run; run += c; }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kXpBatch / kXpThreads; ++i) {
            const uint32_t slot = q0 + (uint32_t)(w * (kXpBatch / kXpWaves) + i * 64 + lane);
            const bool live = slot < E;
            const uint32_t d = dig[i];
            // stable rank inside the (wave, digit) run: one LDS atomic, or the match-any ballots (common.h take_run_slot)
            const uint32_t pos = take_run_slot<kAtomicRank>(s_count[w], d, live, kBits);
            if (live) { s_id[pos] = id[i]; s_dig[pos] = (uint16_t)d; if (AXIS == 0) s_rows[pos] = rows[i]; }
        }
        __syncthreads();
        // coalesced write-out: consecutive batch slots of one digit are consecutive in global memory
        const uint32_t count = min((uint32_t)kXpBatch, E - q0);
#pragma unroll
        for (int i = 0; i < kXpBatch / kXpThreads; ++i) {
            const uint32_t li = (uint32_t)(i * kXpThreads + tid);
            if (li < count) {
                const uint32_t d = s_dig[li];
                const uint32_t g = s_goff[d] + (li - s_lstart[d]);
                if (AXIS == 0) columns_out[g] = make_uint2(s_id[li], s_rows[li] | (d << 21));
                else point_list[g] = s_id[li];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPer; ++j) { const int d = tid * kPer + j; if (d < kBins) s_goff[d] += tot[j]; }
    }
}

// K5: tile ranges = exclusive scan of the per-tile duplicate counts (accumulated by the last pass of the tile partition), and the
// dispatch order of the blend waves: tiles by descending power-of-two class of their list length, index order inside a class
// (stable).  One wave per tile makes the longest lists the tail of K6/K7; starting them first shortens it on scenes with
// heavy-tailed tile loads, and on uniform scenes (one or two classes) the order stays the locality-friendly index order.
// One block: thread t owns a contiguous chunk of tiles.
constexpr int kOrderThreads = 1024, kOrderClasses = 16;
// inclusive prefix sum inside every row of 16 lanes (DPP row shifts)
__device__ __forceinline__ uint32_t row_inclusive_scan16(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    return (uint32_t)v;
}
__global__ __launch_bounds__(kOrderThreads) void tile_ranges_order_kernel(int n_tiles, const uint32_t* __restrict__ counts, uint2* __restrict__ ranges,
                                                                          uint32_t* __restrict__ order) {
    constexpr int kPer = 8, kWaves = kOrderThreads / 64, kChunk = kOrderThreads * kPer;
    static_assert(kWaves == 16 && kOrderClasses == 16, "the cross-wave prefix below runs on 16 x 16 values, one 16-lane row per class");
    __shared__ uint32_t s_cls[kOrderClasses][kWaves];   // tiles per (class, wave) of the current chunk
    __shared__ uint32_t s_off[kOrderClasses][kWaves];   // ... of the lower waves
    __shared__ uint32_t s_tot[kOrderClasses];           // tiles per class of the current chunk
    __shared__ uint32_t s_wsum[kWaves], s_woff[kWaves];
    __shared__ uint32_t s_carry, s_cls_base[kOrderClasses];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // class 0: 2^20 entries and more, ..., class 14: 64..127, class 15: fewer than 64 (or none)
    auto cls_of = [](uint32_t len) { return len ? min(kOrderClasses - 1, max(0, (int)__clz(len) - 11)) : kOrderClasses - 1; };
    const bool single_chunk = n_tiles <= kChunk;   // (every tile shape at 1920 x 1080 but 8 x 8): the class totals fall out of the chunk itself
    if (tid < kOrderClasses) s_cls_base[tid] = 0;
    if (tid == 0) s_carry = 0;
    if (!single_chunk) {
        // class totals over all tiles first (coalesced, one tile per thread and trip): where every class starts in `order`
        uint32_t my_cls[kOrderClasses];
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) my_cls[k] = 0;
        for (int t = tid; t < n_tiles; t += kOrderThreads) { const int c = cls_of(counts[t]);
#pragma unroll
            for (int k = 0; k < kOrderClasses; ++k) my_cls[k] += (c == k); }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) {
            const uint32_t v = wave_inclusive_scan(my_cls[k]);   // lane 63: the wave's total
            if (lane == 63 && v) atomicAdd(&s_cls_base[k], v);
        }
        __syncthreads();
        if (tid == 0) { uint32_t run = 0; for (int k = 0; k < kOrderClasses; ++k) { const uint32_t c = s_cls_base[k]; s_cls_base[k] = run; run += c; } }
    }
    __syncthreads();
    // chunks of kOrderThreads * kPer consecutive tiles, thread t owns kPer consecutive ones: ranges by a block scan with carry,
    // order positions by a per-class scan in tile order (stable inside a class)
    for (int base = 0; base < n_tiles; base += kChunk) {
        const int t0 = base + tid * kPer;
        uint32_t c[kPer], sum = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) { c[i] = t0 + i < n_tiles ? counts[t0 + i] : 0u; sum += c[i]; }
        const uint32_t incl = wave_inclusive_scan(sum);
        if (lane == 63) s_wsum[w] = incl;
        // tiles per class in this thread / wave
        uint32_t mine[kOrderClasses];
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) mine[k] = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) { if (t0 + i < n_tiles) { const int cl = cls_of(c[i]);
#pragma unroll
            for (int k = 0; k < kOrderClasses; ++k) mine[k] += (cl == k); } }
        uint32_t before[kOrderClasses];   // tiles of class k in lower lanes of this wave
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) {
            const uint32_t v = wave_inclusive_scan(mine[k]);
            before[k] = v - mine[k];
            if (lane == 63) s_cls[k][w] = v;
        }
        __syncthreads();
        // prefix over the waves: 16 classes x 16 waves on 256 threads (a 16-lane row per class), the wave totals on 16 more
        if (tid < kOrderClasses * kWaves) {
            const uint32_t v = s_cls[tid >> 4][tid & 15], in = row_inclusive_scan16(v);
            s_off[tid >> 4][tid & 15] = in - v;
            if ((tid & 15) == 15) s_tot[tid >> 4] = in;
        } else if (tid < kOrderClasses * kWaves + 64) {
            const int j = tid - kOrderClasses * kWaves;
            const uint32_t v = j < kWaves ? s_wsum[j] : 0u, in = row_inclusive_scan16(v);
            if (j < kWaves) s_woff[j] = in - v;
        }
        __syncthreads();
        if (single_chunk) {
            if (tid == 0) { uint32_t run = 0; for (int k = 0; k < kOrderClasses; ++k) { s_cls_base[k] = run; run += s_tot[k]; } }
            __syncthreads();
        }
        uint32_t run = s_carry + s_woff[w] + incl - sum;
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) before[k] += s_cls_base[k] + s_off[k][w];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (t0 + i < n_tiles) {
                ranges[t0 + i] = c[i] ? make_uint2(run, run + c[i]) : make_uint2(0u, 0u);
                run += c[i];
                const int cl = cls_of(c[i]);
                uint32_t pos = 0;
#pragma unroll
                for (int k = 0; k < kOrderClasses; ++k) { if (cl == k) { pos = before[k]; before[k] += 1; } }
                order[pos] = (uint32_t)(t0 + i);
            }
        }
        __syncthreads();
        if (tid == kOrderThreads - 1) s_carry = run;
        if (tid < kOrderClasses) s_cls_base[tid] += s_tot[tid];
        __syncthreads();
    }
}

// temp-storage sizes (pure host arithmetic) --------------------------------------------------------
size_t depth_sort_temp_bytes(int P) {
    const uint32_t n = (uint32_t)(P > 0 ? P : 1);
    return align_up(radix_sort_temp_bytes(n), 256);
}
size_t tile_scan_temp_bytes(int P) { return tile_count_scan_temp_bytes((uint32_t)(P > 0 ? P : 1)); }

// histogram tables of the two passes: [digits][blocks] words.  Pass X's lives in the depth sort's scratch (free by then, and
// always large enough: 8 B per Gaussian against <= 4 B), pass Y's in the binning buffer.
size_t expand_x_hist_bytes(int P, int tiles_x) { return (size_t)tiles_x * (((size_t)(P > 0 ? P : 1) + kXpInputsX - 1) / kXpInputsX) * 4; }
size_t expand_y_hist_bytes(uint32_t D, int tiles_y) { return (size_t)tiles_y * (((size_t)(D > 0 ? D : 1) + kXpInputsY - 1) / kXpInputsY) * 4; }

// launchers ---------------------------------------------------------------------------------------
// K2: stable sort of (depth key, gaussian id) -- ties keep ascending id, culled Gaussians (key 0xFFFFFFFF) are dropped.
// `n_visible` (device word, the emission scan's count of Gaussians with a tile): the sort compacts -- only the first *n_visible entries of
// the three outputs are written, the culled Gaussians are dropped by the first pass (radix_sort.hip).
hipError_t run_depth_sort(int P, const uint32_t* depth_keys, const uint2* rect, uint32_t* sorted_keys,
                          uint32_t* sorted_gid, uint2* rect_sorted, void* temp, size_t temp_bytes, int rank_mode, int tiles_x, int tiles_y,
                          const uint32_t* n_visible, hipStream_t s) {
    if (P == 0) return hipSuccess;
    // the sort also delivers the tile rectangles in depth order, so the scan and the partition read sequentially: packed into a word that
    // rides along with the Gaussian's id where its fields (0 .. tiles_x, 0 .. tiles_y) fit into 32 bits -- up to 255 x 255 tiles, or
    // e.g. 511 x 127 -- and gathered by the last pass for wider frames (radix_sort.hip)
    const int bx = 32 - __builtin_clz((unsigned)(tiles_x > 0 ? tiles_x : 1)), by = 32 - __builtin_clz((unsigned)(tiles_y > 0 ? tiles_y : 1));
    return radix_sort_pairs(depth_keys, nullptr, sorted_keys, sorted_gid, (uint32_t)P, 32, temp, temp_bytes, s, rect, rect_sorted, rank_mode, bx, by, n_visible);
}

// K2: emission offsets = scan of tiles_touched in id order (block-local values in first, block bases + total D in block_base).
hipError_t run_tile_count_scan(int P, const uint32_t* tiles_touched, uint32_t* first, void* block_base, size_t base_bytes, uint32_t* total_host,
                               hipStream_t s) {
    if (P == 0) return hipSuccess;
    return tile_count_scan(tiles_touched, first, (uint32_t)P, block_base, base_bytes, total_host, s);
}

template <int AXIS, bool kAtomicRank, typename... Args>
static void launch_expand_scatter_r(int bins, int blocks, hipStream_t s, Args... a) {
    if (bins <= 128) hipLaunchKernelGGL((expand_scatter_kernel<AXIS, 7, kAtomicRank>), dim3(blocks), dim3(kXpThreads), 0, s, a...);
    else if (bins <= 256) hipLaunchKernelGGL((expand_scatter_kernel<AXIS, 8, kAtomicRank>), dim3(blocks), dim3(kXpThreads), 0, s, a...);
    else if (bins <= 512) hipLaunchKernelGGL((expand_scatter_kernel<AXIS, 9, kAtomicRank>), dim3(blocks), dim3(kXpThreads), 0, s, a...);
    else hipLaunchKernelGGL((expand_scatter_kernel<AXIS, 10, kAtomicRank>), dim3(blocks), dim3(kXpThreads), 0, s, a...);
}
template <int AXIS, typename... Args>
static void launch_expand_scatter(int rank_mode, int bins, int blocks, hipStream_t s, Args... a) {
    if (rank_mode == kRankAtomic) launch_expand_scatter_r<AXIS, true>(bins, blocks, s, a...);
    else launch_expand_scatter_r<AXIS, false>(bins, blocks, s, a...);
}

// K3: Gaussians in depth order -> column items ordered by (tile column, depth).  Also zeroes tile_counts.  n_columns (device word) receives the number of column items.
hipError_t run_expand_columns(int P, int tiles_x, int n_tiles, const uint2* rect_sorted, const uint32_t* sorted_gid, uint2* columns,
                              uint32_t* n_columns, uint32_t* hist, uint32_t* row_total, uint32_t* tile_counts, int rank_mode, const uint32_t* n_visible,
                              hipStream_t s) {
    if (P == 0) return hipSuccess;
    if (tiles_x > kXpMaxBins || (rank_mode != kRankAtomic && rank_mode != kRankBallot)) return hipErrorInvalidValue;
    const int nb = (P + kXpInputsX - 1) / kXpInputsX;
    // (n_visible: the depth sort compacted -- only the first *n_visible sorted entries exist; the grid is sized for P, surplus blocks leave at once)
    hipLaunchKernelGGL(expand_hist_kernel<0>, dim3(nb), dim3(kXpThreads), 0, s, rect_sorted, (uint32_t)P, n_visible, tiles_x, hist, nb,
                       tiles_x, tile_counts, n_tiles);
    hipLaunchKernelGGL(expand_scan_rows_kernel, dim3(tiles_x), dim3(kXpThreads), 0, s, hist, nb, (uint32_t)P, n_visible, kXpInputsX, row_total);
    launch_expand_scatter<0>(rank_mode, tiles_x, nb, s, rect_sorted, (uint32_t)P, n_visible, tiles_x, (const uint32_t*)hist, nb,
                             (const uint32_t*)row_total, sorted_gid, columns, n_columns, (uint32_t*)nullptr);
    return hipGetLastError();
}

// K4: column items -> point_list ordered by (tile row, tile column, depth) + the entries per tile.  The number of column items is
// only known on the device; the grid is sized for the upper bound D and surplus blocks leave at once.
hipError_t run_expand_rows(uint32_t D, int tiles_x, int tiles_y, const uint2* columns, const uint32_t* n_columns, uint32_t* hist, uint32_t* row_total,
                           uint32_t* point_list, uint32_t* tile_counts, int rank_mode, hipStream_t s) {
    if (D == 0) return hipSuccess;
    if (tiles_y > kXpMaxBins || (rank_mode != kRankAtomic && rank_mode != kRankBallot)) return hipErrorInvalidValue;
    const int nb = (int)((D + kXpInputsY - 1) / kXpInputsY);
    hipLaunchKernelGGL(expand_hist_kernel<1>, dim3(nb), dim3(kXpThreads), 0, s, columns, D, n_columns, tiles_y, hist, nb, tiles_x, tile_counts,
                       tiles_x * tiles_y);
    hipLaunchKernelGGL(expand_scan_rows_kernel, dim3(tiles_y), dim3(kXpThreads), 0, s, hist, nb, D, n_columns, kXpInputsY, row_total);
    launch_expand_scatter<1>(rank_mode, tiles_y, nb, s, columns, D, n_columns, tiles_y, (const uint32_t*)hist, nb, (const uint32_t*)row_total,
                             (const uint32_t*)nullptr, (uint2*)nullptr, (uint32_t*)nullptr, point_list);
    return hipGetLastError();
}

// SR_FLAG_BINNING_CAPACITY (the forward without its host read-back): the caller sized the binning buffer from a GUESS of the frame's
// duplicate count.  counts = [D, visible Gaussians, overflow] in the geometry state (the emission scan's totals; the third word was a
// per-block count the scan is done with).  If the frame's D exceeds the capacity, the visible count is zeroed -- pass X takes its item
// count from that word, so nothing is emitted, no list is built, the blend writes the background -- and the overflow word is set: the
// caller finds out when it next looks (sr_geom_view: frame_counts[2]) and renders that frame again with a buffer of D items.
__global__ void capacity_guard_kernel(uint32_t* __restrict__ counts, uint32_t capacity) {
    const uint32_t over = counts[0] > capacity ? 1u : 0u;
    counts[2] = over;
    if (over) counts[1] = 0u;
}
// Zero fill by a kernel of the library's own (the `written` flags of the backward, the tile counters of an empty frame).  Not
// hipMemsetAsync: captured into a HIP graph, its memset node did not clear a 21 499-byte span on replay (ROCm 7.2: stale `written`
// flags of the previous replay reached K8 -- tests/test_gpu_boundary.py, the whole-step graph test, replays on CHANGED inputs to see it).
__global__ __launch_bounds__(256) void zero_bytes_kernel(uint8_t* __restrict__ p, size_t n) {
    // head up to 16-B alignment and the tail by bytes, the body by 16-B stores
    const size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15);
    const size_t h = head < n ? head : n;
    const size_t body = (n - h) / 16;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    for (size_t k = i; k < body; k += stride) q[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i < h) p[i] = 0;
    const size_t tail0 = h + body * 16;
    if (i < n - tail0) p[tail0 + i] = 0;
}
hipError_t launch_zero_bytes(void* p, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t blocks = (n / 16 + 255) / 256;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks))), dim3(256), 0, s, static_cast<uint8_t*>(p), n);
    return hipGetLastError();
}

hipError_t run_capacity_guard(uint32_t* counts, uint32_t capacity, hipStream_t s) {
    hipLaunchKernelGGL(capacity_guard_kernel, dim3(1), dim3(1), 0, s, counts, capacity);
    return hipGetLastError();
}

hipError_t run_tile_ranges_order(int n_tiles, const uint32_t* tile_counts, uint2* ranges, uint32_t* order, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_ranges_order_kernel, dim3(1), dim3(kOrderThreads), 0, s, n_tiles, tile_counts, ranges, order);
    return hipGetLastError();
}

}  // namespace sr
