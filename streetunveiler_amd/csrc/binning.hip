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
//      of 8-byte pairs instead of 6 passes of 12-byte pairs).
// Stability of both sorts makes (tile, depth, id) the final order.  Integer work only.
#include "common.h"

namespace sr {

// radix_sort.hip
size_t radix_sort_temp_bytes(uint32_t n);
hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint32_t* aux_src, uint32_t* aux_out);
size_t gather_scan_temp_bytes(uint32_t n);
hipError_t gather_inclusive_scan(const uint32_t* idx, const uint32_t* src, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes,
                                 hipStream_t s);

// K3: wave-cooperative duplicate emission.  Each wave owns 64 consecutive depth ranks; the lanes
// then walk the wave's contiguous output span 64 slots at a time (coalesced 4-B stores), finding the
// owning Gaussian of each slot by a 6-step binary search over the wave's exclusive offsets in LDS.
__global__ __launch_bounds__(256) void emit_duplicates_kernel(int P, int tiles_x, const uint2* __restrict__ rect,
                                                              const uint32_t* __restrict__ sorted_gid,
                                                              const uint32_t* __restrict__ sorted_offsets,
                                                              float4* __restrict__ recs,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_start[4][65];
    __shared__ uint32_t s_gid[4][64];
    __shared__ int s_minx[4][64], s_miny[4][64], s_w[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t incl = 0, count = 0, gid = 0;
    int minx = 0, miny = 0, w = 0;
    if (r < P) {
        gid = sorted_gid[r];
        incl = sorted_offsets[r];
        const uint32_t prev = r > 0 ? sorted_offsets[r - 1] : 0u;
        count = incl - prev;
        if (count) {   // K1's tile rectangle, one 8-B gather instead of two record quads
            const uint2 rc = rect[gid];
            minx = (int)(rc.x & 0xFFFFu); miny = (int)(rc.x >> 16); w = (int)(rc.y & 0xFFFFu);
        }
    } else {
        // ranks past P: inherit the last inclusive offset so the search stays monotone
        incl = sorted_offsets[P - 1];
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

// K5: tile ranges from the sorted tile keys.
__global__ void tile_ranges_kernel(uint32_t D, const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const uint32_t t = tile_keys[j];
    if (j == 0) ranges[t].x = 0;
    else {
        const uint32_t p = tile_keys[j - 1];
        if (p != t) { ranges[p].y = j; ranges[t].x = j; }
    }
    if (j == D - 1) ranges[t].y = D;
}

// Dispatch order of the blend waves: tiles by descending power-of-two class of their list length, index order inside a class
// (stable).  One wave per tile makes the longest lists the tail of K6/K7; starting them first shortens it on scenes with
// heavy-tailed tile loads, and on uniform scenes (one or two classes) the order stays the locality-friendly index order.
constexpr int kOrderThreads = 512, kOrderClasses = 16;
__global__ __launch_bounds__(kOrderThreads) void tile_order_kernel(int n_tiles, const uint2* __restrict__ ranges, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_cnt[kOrderClasses * kOrderThreads];   // [class, longest first][thread]: the linear index IS the output order
    __shared__ uint32_t s_wsum[kOrderThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chunk = (n_tiles + kOrderThreads - 1) / kOrderThreads, begin = min(n_tiles, tid * chunk), end = min(n_tiles, begin + chunk);
    // class 0: 2^20 entries and more, ..., class 14: 64..127, class 15: fewer than 64 (or none)
    auto row = [](uint2 r) { const uint32_t len = r.y - r.x; return len ? min(kOrderClasses - 1, max(0, (int)__clz(len) - 11)) : kOrderClasses - 1; };
    for (int k = 0; k < kOrderClasses; ++k) s_cnt[k * kOrderThreads + tid] = 0;
    for (int t = begin; t < end; ++t) s_cnt[row(ranges[t]) * kOrderThreads + tid] += 1;   // own column: no conflicts
    __syncthreads();
    // exclusive scan of the counters in linear order: thread t owns the kOrderClasses consecutive ones starting at kOrderClasses * t
    uint32_t local[kOrderClasses], sum = 0;
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) { local[k] = s_cnt[tid * kOrderClasses + k]; sum += local[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    uint32_t off = incl - sum;
    for (int j = 0; j < w; ++j) off += s_wsum[j];
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) { s_cnt[tid * kOrderClasses + k] = off; off += local[k]; }
    __syncthreads();
    for (int t = begin; t < end; ++t) { const int k = row(ranges[t]); order[s_cnt[k * kOrderThreads + tid]++] = (uint32_t)t; }
}

static int bits_for(uint32_t n) {  // number of bits needed to represent values in [0, n)
    int b = 0;
    while ((1ull << b) < (unsigned long long)n) ++b;
    return b < 1 ? 1 : b;
}

// temp-storage sizes (pure host arithmetic) --------------------------------------------------------
size_t depth_sort_temp_bytes(int P) {
    const uint32_t n = (uint32_t)(P > 0 ? P : 1);
    const size_t a = radix_sort_temp_bytes(n), b = gather_scan_temp_bytes(n);
    return align_up(a > b ? a : b, 256);
}

size_t tile_sort_temp_bytes(uint32_t D, int n_tiles) {
    (void)n_tiles;
    return align_up(radix_sort_temp_bytes(D > 0 ? D : 1), 256);
}

// launchers ---------------------------------------------------------------------------------------
// K2: stable sort of (depth key, gaussian id) -- ties keep ascending id, culled Gaussians (key 0xFFFFFFFF) end up last.
hipError_t run_depth_sort(int P, const uint32_t* depth_keys, const uint32_t* tiles_touched, uint32_t* sorted_keys,
                          uint32_t* sorted_gid, uint32_t* tt_sorted, void* temp, size_t temp_bytes, hipStream_t s) {
    if (P == 0) return hipSuccess;
    // the last pass also gathers tiles_touched into depth order, so the scan reads sequentially
    return radix_sort_pairs(depth_keys, nullptr, sorted_keys, sorted_gid, (uint32_t)P, 32, temp, temp_bytes, s, tiles_touched, tt_sorted);
}

// K2: inclusive scan of the tile counts in depth order -> where each Gaussian's duplicates end in emission order.
hipError_t run_tile_count_scan(int P, const uint32_t* tt_sorted, uint32_t* sorted_offsets, void* temp, size_t temp_bytes, hipStream_t s) {
    if (P == 0) return hipSuccess;
    return gather_inclusive_scan(nullptr, tt_sorted, sorted_offsets, (uint32_t)P, temp, temp_bytes, s);
}

hipError_t run_emit(int P, int tiles_x, const uint2* rect, const uint32_t* sorted_gid, const uint32_t* sorted_offsets,
                    float4* recs, uint32_t* keys_unsorted, uint32_t* vals_unsorted, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_duplicates_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, tiles_x, rect, sorted_gid,
                       sorted_offsets, recs, keys_unsorted, vals_unsorted);
    return hipGetLastError();
}

hipError_t run_tile_sort(uint32_t D, int n_tiles, const uint32_t* keys_unsorted, const uint32_t* vals_unsorted,
                         uint32_t* tile_keys, uint32_t* point_list, void* temp, size_t temp_bytes, hipStream_t s) {
    if (D == 0) return hipSuccess;
    return radix_sort_pairs(keys_unsorted, vals_unsorted, tile_keys, point_list, D, bits_for((uint32_t)n_tiles), temp, temp_bytes, s, nullptr,
                            nullptr);
}

hipError_t run_tile_ranges(uint32_t D, int n_tiles, const uint32_t* tile_keys, uint2* ranges, hipStream_t s) {
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)n_tiles, s);
    if (e != hipSuccess || D == 0) return e;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((D + 255) / 256), dim3(256), 0, s, D, tile_keys, ranges);
    return hipGetLastError();
}

hipError_t run_tile_order(int n_tiles, const uint2* ranges, uint32_t* order, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(kOrderThreads), 0, s, n_tiles, ranges, order);
    return hipGetLastError();
}

}  // namespace sr
