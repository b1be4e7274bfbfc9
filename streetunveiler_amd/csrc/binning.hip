// binning.hip -- K2..K5: depth ordering of Gaussians, tile-count scan, duplicate emission,
// stable tile partition, tile ranges.
//
// The reference algorithm emits one (tile<<32 | depth_bits, gaussian)
// pair per touched tile and LSD-radix-sorts all D 64-bit keys.  The result is the list ordered by
// (tile, depth bits, gaussian id).  This build produces the *same list, bit for bit*, with ~1/3 of
// the HBM traffic by splitting the key:
//   1. sort the P Gaussians once by their 32-bit depth key (stable => ties keep ascending id);
//   2. emit duplicates in that order (coalesced, wave-cooperative); the duplicates of one Gaussian are
//      contiguous in this emission order (first[gid] + (ty - miny) * w + (tx - minx)), which is where
//      K7 stores the per-duplicate gradient records so that K8 can sum them as one contiguous span;
//   3. stable-partition the D duplicates by tile id only (ceil(log2(tiles)) bits, 2 radix passes
//      of 8-byte pairs instead of 6 passes of 12-byte pairs); the last pass keeps only the permutation
//      (point_list) and counts the duplicates per tile on the way, so the tile ranges are one exclusive
//      scan of 8 160 counters -- the sorted keys are never written or read back.
// Stability of both sorts makes (tile, depth, id) the final order.  Integer work only.
#include "common.h"

namespace sr {

// radix_sort.hip
size_t radix_sort_temp_bytes(uint32_t n);
hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint2* aux_src, uint2* aux_out,
                            uint32_t* full_hist);
size_t tile_count_scan_temp_bytes(uint32_t n);
hipError_t tile_count_scan(const uint2* rect_sorted, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes, hipStream_t s);
constexpr int kScanTile = 2048;   // ranks per block of the tile-count scan (radix_sort.hip kRsTile)

// K3: wave-cooperative duplicate emission.  Each wave owns 64 consecutive depth ranks; the lanes
// then walk the wave's contiguous output span 64 slots at a time (coalesced 4-B stores), finding the
// owning Gaussian of each slot by a 6-step binary search over the wave's exclusive offsets in LDS.
__global__ __launch_bounds__(256) void emit_duplicates_kernel(int P, int tiles_x, const uint2* __restrict__ rect_sorted,
                                                              const uint32_t* __restrict__ sorted_gid,
                                                              const uint32_t* __restrict__ block_offsets,   // inclusive scan inside blocks of kScanTile ranks
                                                              const uint32_t* __restrict__ block_base,      // exclusive scan of the block totals; [nblocks] = D
                                                              float4* __restrict__ recs,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out,
                                                              uint32_t* __restrict__ tile_counts, int n_tiles) {
    __shared__ uint32_t s_start[4][65];
    __shared__ uint32_t s_gid[4][64];
    __shared__ int s_minx[4][64], s_miny[4][64], s_w[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    // side job: clear the per-tile duplicate counters that the last pass of the tile partition accumulates into
    for (int k = r; k < n_tiles; k += gridDim.x * blockDim.x) tile_counts[k] = 0;
    // global inclusive offset of a rank = its block-local scan value + the base of its scan block
    auto offset_of = [&](int rank) { return block_offsets[rank] + block_base[rank / kScanTile]; };
    uint32_t incl = 0, count = 0, gid = 0;
    int minx = 0, miny = 0, w = 0;
    if (r < P) {
        gid = sorted_gid[r];
        const uint2 rc = rect_sorted[r];   // K1's tile rectangle, gathered into depth order by the last pass of the depth sort
        minx = (int)(rc.x & 0xFFFFu); miny = (int)(rc.x >> 16); w = (int)(rc.y & 0xFFFFu);
        count = (uint32_t)w * (rc.y >> 16);
        incl = offset_of(r);
    } else {
        // ranks past P: inherit the last inclusive offset so the search stays monotone
        incl = offset_of(P - 1);
    }
    const uint32_t excl = incl - count;
    // emission index of this Gaussian's first duplicate, parked in slot 15 of its record (K7 derives the others and gets
    // it for free with the record gather)
    if (r < P && count) reinterpret_cast<float*>(recs)[(size_t)gid * kRecFloats + 15] = __uint_as_float(excl);
    s_start[wave][lane] = excl;
    if (lane == 63) s_start[wave][64] = incl;
    s_gid[wave][lane] = gid; s_minx[wave][lane] = minx; s_miny[wave][lane] = miny; s_w[wave][lane] = w;
    __syncthreads();
    const uint32_t span_begin = s_start[wave][0], span_end = s_start[wave][64];
    for (uint32_t slot = span_begin + lane; slot < span_end; slot += 64) {
        // largest j with s_start[j] <= slot  (counts of zero are skipped automatically)
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (s_start[wave][lo + step] <= slot) lo += step;
        const uint32_t local = slot - s_start[wave][lo];
        const int ww = s_w[wave][lo];
        const int dy = (int)(local / (uint32_t)ww), dx = (int)(local - (uint32_t)dy * (uint32_t)ww);
        keys_out[slot] = (uint32_t)((s_miny[wave][lo] + dy) * tiles_x + s_minx[wave][lo] + dx);
        vals_out[slot] = s_gid[wave][lo];
    }
}

// K5: tile ranges = exclusive scan of the per-tile duplicate counts (accumulated by the last pass of the tile partition), and the
// dispatch order of the blend waves: tiles by descending power-of-two class of their list length, index order inside a class
// (stable).  One wave per tile makes the longest lists the tail of K6/K7; starting them first shortens it on scenes with
// heavy-tailed tile loads, and on uniform scenes (one or two classes) the order stays the locality-friendly index order.
// One block: thread t owns a contiguous chunk of tiles.
constexpr int kOrderThreads = 1024, kOrderClasses = 16;
__global__ __launch_bounds__(kOrderThreads) void tile_ranges_order_kernel(int n_tiles, const uint32_t* __restrict__ counts, uint2* __restrict__ ranges,
                                                                          uint32_t* __restrict__ order) {
    __shared__ uint32_t s_cls[kOrderClasses][kOrderThreads / 64];   // tiles per (class, wave) -> order offsets
    __shared__ uint32_t s_wsum[kOrderThreads / 64];
    __shared__ uint32_t s_carry, s_cls_base[kOrderClasses];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int kPer = 8, kWaves = kOrderThreads / 64;
    // class 0: 2^20 entries and more, ..., class 14: 64..127, class 15: fewer than 64 (or none)
    auto cls_of = [](uint32_t len) { return len ? min(kOrderClasses - 1, max(0, (int)__clz(len) - 11)) : kOrderClasses - 1; };
    // pass 1: class totals over all tiles (coalesced, one tile per thread and trip)
    uint32_t my_cls[kOrderClasses];
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) my_cls[k] = 0;
    for (int t = tid; t < n_tiles; t += kOrderThreads) { const int c = cls_of(counts[t]);
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) my_cls[k] += (c == k); }
    if (tid < kOrderClasses) s_cls_base[tid] = 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) {
        uint32_t v = my_cls[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
        if (lane == 0 && v) atomicAdd(&s_cls_base[k], v);
    }
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (int k = 0; k < kOrderClasses; ++k) { const uint32_t c = s_cls_base[k]; s_cls_base[k] = run; run += c; } }
    __syncthreads();
    // pass 2: chunks of kOrderThreads * kPer consecutive tiles, thread t owns kPer consecutive ones: ranges by a block scan with carry,
    // order positions by a per-class scan in tile order (stable inside a class)
    for (int base = 0; base < n_tiles; base += kOrderThreads * kPer) {
        const int t0 = base + tid * kPer;
        uint32_t c[kPer], sum = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) { c[i] = t0 + i < n_tiles ? counts[t0 + i] : 0u; sum += c[i]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += y; }
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
            uint32_t v = mine[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += y; }
            before[k] = v - mine[k];
            if (lane == 63) s_cls[k][w] = v;
        }
        __syncthreads();
        uint32_t run = s_carry + incl - sum;
        for (int j = 0; j < w; ++j) run += s_wsum[j];
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) { uint32_t o = s_cls_base[k] + before[k]; for (int j = 0; j < w; ++j) o += s_cls[k][j]; before[k] = o; }
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
        if (tid < kOrderClasses) { uint32_t add = 0; for (int j = 0; j < kWaves; ++j) add += s_cls[tid][j]; s_cls_base[tid] += add; }
        __syncthreads();
    }
}

static int bits_for(uint32_t n) {  // number of bits needed to represent values in [0, n)
    int b = 0;
    while ((1ull << b) < (unsigned long long)n) ++b;
    return b < 1 ? 1 : b;
}

// temp-storage sizes (pure host arithmetic) --------------------------------------------------------
size_t depth_sort_temp_bytes(int P) {
    const uint32_t n = (uint32_t)(P > 0 ? P : 1);
    return align_up(radix_sort_temp_bytes(n), 256);
}
size_t tile_scan_temp_bytes(int P) { return tile_count_scan_temp_bytes((uint32_t)(P > 0 ? P : 1)); }

size_t tile_sort_temp_bytes(uint32_t D, int n_tiles) {
    (void)n_tiles;
    return align_up(radix_sort_temp_bytes(D > 0 ? D : 1), 256);
}

// launchers ---------------------------------------------------------------------------------------
// K2: stable sort of (depth key, gaussian id) -- ties keep ascending id, culled Gaussians (key 0xFFFFFFFF) end up last.
hipError_t run_depth_sort(int P, const uint32_t* depth_keys, const uint2* rect, uint32_t* sorted_keys,
                          uint32_t* sorted_gid, uint2* rect_sorted, void* temp, size_t temp_bytes, hipStream_t s) {
    if (P == 0) return hipSuccess;
    // the last pass also gathers the tile rectangles into depth order, so the scan and the emission read sequentially
    return radix_sort_pairs(depth_keys, nullptr, sorted_keys, sorted_gid, (uint32_t)P, 32, temp, temp_bytes, s, rect, rect_sorted, nullptr);
}

// K2: scan of the tile counts in depth order (block-local values in block_offsets, block bases + total in block_base).
hipError_t run_tile_count_scan(int P, const uint2* rect_sorted, uint32_t* block_offsets, void* block_base, size_t base_bytes, hipStream_t s) {
    if (P == 0) return hipSuccess;
    return tile_count_scan(rect_sorted, block_offsets, (uint32_t)P, block_base, base_bytes, s);
}

hipError_t run_emit(int P, int tiles_x, const uint2* rect_sorted, const uint32_t* sorted_gid, const uint32_t* block_offsets,
                    const uint32_t* block_base, float4* recs, uint32_t* keys_unsorted, uint32_t* vals_unsorted, uint32_t* tile_counts,
                    int n_tiles, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_duplicates_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, tiles_x, rect_sorted, sorted_gid,
                       block_offsets, block_base, recs, keys_unsorted, vals_unsorted, tile_counts, n_tiles);
    return hipGetLastError();
}

// K4: only the permutation leaves the last pass; tile_counts[tile] (zeroed by the emission kernel) receives the duplicates per tile
hipError_t run_tile_sort(uint32_t D, int n_tiles, const uint32_t* keys_unsorted, const uint32_t* vals_unsorted,
                         uint32_t* point_list, uint32_t* tile_counts, void* temp, size_t temp_bytes, hipStream_t s) {
    if (D == 0) return hipMemsetAsync(tile_counts, 0, sizeof(uint32_t) * (size_t)n_tiles, s);   // (no emission ran)
    return radix_sort_pairs(keys_unsorted, vals_unsorted, nullptr, point_list, D, bits_for((uint32_t)n_tiles), temp, temp_bytes, s, nullptr,
                            nullptr, tile_counts);
}

hipError_t run_tile_ranges_order(int n_tiles, const uint32_t* tile_counts, uint2* ranges, uint32_t* order, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_ranges_order_kernel, dim3(1), dim3(kOrderThreads), 0, s, n_tiles, tile_counts, ranges, order);
    return hipGetLastError();
}

}  // namespace sr
