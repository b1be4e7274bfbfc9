// K-nearest-neighbour mean squared distance == `dist3knn` / `dist10knn` / `meanDistFromReferencePcd` of the
// reference's simple-knn fork [REF /root/reference/scene/gaussian_model.py:16,151; scene/mask_gaussian.py:21;
// inpainting_pipeline/2_condition_preparation/2_generate_inpainted_mask.py:27,71-73].  The fork's source is an un-vendored
// submodule (/root/reference/.gitmodules: submodules/simple-knn, no pinned SHA); what is restated is the published
// simple-knn algorithm it descends from: out[i] = mean of the K smallest SQUARED distances from point i to the other points.
//
// The search is exact.  Points are ordered along a 30-bit Morton curve (hand-written radix sort, radix_sort.hip), cut into
// boxes of 512 consecutive points with their AABBs, and one wave64 handles 64 curve-consecutive queries: lanes test 64 boxes
// at a time against the wave's query AABB and its current worst K-th distance (ballot), and every surviving box is scanned
// by all lanes together -- the candidate is wave-uniform (scalar loads), the distance and the branch-free K-best insertion
// run per lane, so there is no divergence and no LDS.  Integer/bandwidth/VALU work only; no MFMA.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "common.h"

namespace sr {

hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint2* aux_src, uint2* aux_out, int rank_mode,
                            int rect_bx = 0, int rect_by = 0, const uint32_t* n_live = nullptr);
size_t radix_sort_temp_bytes(uint32_t n);

constexpr int kWave = 64;
constexpr int kKnnBox = 512;
constexpr int kKnnThreads = 256;

// ---- bounding box -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kKnnThreads) void knn_bounds_partial_kernel(const float* __restrict__ pts, int n, float* __restrict__ partial) {
    __shared__ float s_red[6][kKnnThreads / kWave];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * kKnnThreads + threadIdx.x; i < n; i += gridDim.x * kKnnThreads) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = pts[3 * (size_t)i + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = kWave / 2; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_red[c][wave] = lo[c]; s_red[3 + c][wave] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[threadIdx.x][0];
        for (int w = 1; w < kKnnThreads / kWave; ++w) v = threadIdx.x < 3 ? fminf(v, s_red[threadIdx.x][w]) : fmaxf(v, s_red[threadIdx.x][w]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ void knn_bounds_final_kernel(const float* __restrict__ partial, int rows, float* __restrict__ bounds) {
    const int c = threadIdx.x;
    if (c >= 6) return;
    float v = partial[c];
    for (int r = 1; r < rows; ++r) v = c < 3 ? fminf(v, partial[r * 6 + c]) : fmaxf(v, partial[r * 6 + c]);
    bounds[c] = v;
}

// ---- Morton codes -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void knn_morton_kernel(const float* __restrict__ pts, int n, const float* __restrict__ bounds, uint32_t* __restrict__ codes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lo = bounds[c], ext = fmaxf(bounds[3 + c] - lo, 1e-30f);
        const float t = (pts[3 * (size_t)i + c] - lo) / ext * 1023.f;
        q[c] = (uint32_t)fminf(fmaxf(t, 0.f), 1023.f);   // NaN -> 0
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}

// sorted[i] = (x, y, z, original index) of the i-th point along the curve
__global__ void knn_gather_kernel(const float* __restrict__ pts, const uint32_t* __restrict__ order, int n, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], __uint_as_float(j));
}

// AABB of each run of kKnnBox curve-consecutive points: boxes[2b] = (min, -), boxes[2b+1] = (max, -)
__global__ __launch_bounds__(kKnnThreads) void knn_boxes_kernel(const float4* __restrict__ sorted, int n, float4* __restrict__ boxes) {
    __shared__ float s_red[6][kKnnThreads / kWave];
    const int base = blockIdx.x * kKnnBox;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = threadIdx.x; j < kKnnBox && base + j < n; j += kKnnThreads) {
        const float4 p = sorted[base + j];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = kWave / 2; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    }
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_red[c][wave] = lo[c]; s_red[3 + c][wave] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKnnThreads / kWave; ++w) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_red[c][0] = fminf(s_red[c][0], s_red[c][w]); s_red[3 + c][0] = fmaxf(s_red[3 + c][0], s_red[3 + c][w]); }
        }
        boxes[2 * blockIdx.x] = make_float4(s_red[0][0], s_red[1][0], s_red[2][0], 0.f);
        boxes[2 * blockIdx.x + 1] = make_float4(s_red[3][0], s_red[4][0], s_red[5][0], 0.f);
    }
}

// ---- search --------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dist2(const float4 a, const float4 b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return (dx * dx + dy * dy) + dz * dz;
}

template <int K>
__device__ __forceinline__ void insert_best(float (&best)[K], float d) {   // keeps best[] ascending; branch-free
#pragma unroll
    for (int k = 0; k < K; ++k) { const float lo = fminf(best[k], d); d = fmaxf(best[k], d); best[k] = lo; }
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    for (int o = kWave / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// kSelf: queries ARE the reference points (same sorted array); a point is not its own neighbour.
template <int K, bool kSelf>
__global__ __launch_bounds__(kKnnThreads) void knn_search_kernel(const float4* __restrict__ q_sorted, int nq,
                                                                  const float4* __restrict__ r_sorted, const uint32_t* __restrict__ r_codes,
                                                                  const uint32_t* __restrict__ q_codes, int nr,
                                                                  const float4* __restrict__ boxes, int n_boxes, int take_sqrt,
                                                                  float* __restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_base = (blockIdx.x * (kKnnThreads / kWave) + threadIdx.x / kWave) * kWave;
    if (wave_base >= nq) return;
    const int pos = wave_base + lane;
    const bool valid = pos < nq;
    const float4 q = q_sorted[valid ? pos : nq - 1];
    float best[K];
#pragma unroll
    for (int k = 0; k < K; ++k) best[k] = FLT_MAX;

    // a first bound from the neighbours along the curve (K on either side); the candidates are looked at again in the box scan
    int centre = pos;
    if (!kSelf) {   // position of the query's code in the reference order
        const uint32_t code = q_codes[valid ? pos : nq - 1];
        int lo = 0, hi = nr;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (r_codes[mid] < code) lo = mid + 1; else hi = mid; }
        centre = lo;
    }
    for (int o = -K; o <= K; ++o) {
        const int j = centre + o;
        if (j < 0 || j >= nr || (kSelf && o == 0)) continue;
        insert_best<K>(best, dist2(q, r_sorted[j]));
    }
    const float reject = best[K - 1];
#pragma unroll
    for (int k = 0; k < K; ++k) best[k] = FLT_MAX;

    // AABB of the wave's queries
    const float big = FLT_MAX;
    const float wlo[3] = {wave_min(valid ? q.x : big), wave_min(valid ? q.y : big), wave_min(valid ? q.z : big)};
    const float whi[3] = {wave_max(valid ? q.x : -big), wave_max(valid ? q.y : -big), wave_max(valid ? q.z : -big)};

    for (int g = 0; g < n_boxes; g += kWave) {
        // the worst bound any lane still has: a box farther than that from the wave's AABB cannot matter to anyone
        const float bound = wave_max(valid ? fminf(reject, best[K - 1]) : 0.f);
        bool need = false;
        if (g + lane < n_boxes) {
            const float4 bl = boxes[2 * (g + lane)], bh = boxes[2 * (g + lane) + 1];
            const float gx = fmaxf(0.f, fmaxf(bl.x - whi[0], wlo[0] - bh.x));
            const float gy = fmaxf(0.f, fmaxf(bl.y - whi[1], wlo[1] - bh.y));
            const float gz = fmaxf(0.f, fmaxf(bl.z - whi[2], wlo[2] - bh.z));
            need = !(((gx * gx + gy * gy) + gz * gz) > bound);
        }
        unsigned long long todo = ballot64(need);
        while (todo) {
            const int b = g + __builtin_ctzll(todo);
            todo &= todo - 1;
            // per-lane test as in the published algorithm: skip when the box is farther than this lane's bounds
            const float4 bl = boxes[2 * b], bh = boxes[2 * b + 1];
            const float px = fmaxf(0.f, fmaxf(bl.x - q.x, q.x - bh.x)), py = fmaxf(0.f, fmaxf(bl.y - q.y, q.y - bh.y)),
                        pz = fmaxf(0.f, fmaxf(bl.z - q.z, q.z - bh.z));
            const float pd = (px * px + py * py) + pz * pz;
            const bool mine = valid && !(pd > reject) && !(pd > best[K - 1]);
            if (ballot64(mine) == 0) continue;
            const int first = b * kKnnBox, last = min(nr, first + kKnnBox);
            for (int j = first; j < last; ++j) {
                const float4 c = r_sorted[j];   // wave-uniform address
                float d = dist2(q, c);
                if (kSelf && j == pos) d = FLT_MAX;
                if (mine) insert_best<K>(best, d);
            }
        }
    }
    if (valid) {
        float sum = best[0];
#pragma unroll
        for (int k = 1; k < K; ++k) sum += best[k];
        const float mean = sum / (float)K;
        out[__float_as_uint(q.w)] = take_sqrt ? sqrtf(mean) : mean;
    }
}

// ---- host ----------------------------------------------------------------------------------------------------
struct KnnLayout {
    size_t partial, bounds, r_codes, r_codes_sorted, r_order, r_sorted, boxes, q_codes, q_codes_sorted, q_order, q_sorted, sort_temp, total;
    int blocks_r, blocks_q, n_boxes;
};

static int bounds_blocks(int n) { const int b = (n + kKnnThreads * 8 - 1) / (kKnnThreads * 8); return b < 1 ? 1 : (b > 1024 ? 1024 : b); }

KnnLayout knn_layout(int nq, int nr) {   // nq == 0: self search
    KnnLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes ? bytes : 1, 256); return o; };
    L.blocks_r = bounds_blocks(nr); L.blocks_q = nq > 0 ? bounds_blocks(nq) : 0;
    L.n_boxes = (nr + kKnnBox - 1) / kKnnBox;
    L.partial = take((size_t)(L.blocks_r + L.blocks_q) * 6 * 4);
    L.bounds = take(6 * 4);
    L.r_codes = take((size_t)nr * 4); L.r_codes_sorted = take((size_t)nr * 4); L.r_order = take((size_t)nr * 4);
    L.r_sorted = take((size_t)nr * 16);
    L.boxes = take((size_t)L.n_boxes * 32);
    L.q_codes = take((size_t)nq * 4); L.q_codes_sorted = take((size_t)nq * 4); L.q_order = take((size_t)nq * 4);
    L.q_sorted = take((size_t)nq * 16);
    L.sort_temp = take(radix_sort_temp_bytes((uint32_t)(nq > nr ? nq : nr)));
    L.total = off;
    return L;
}

size_t knn_workspace_bytes(int nq, int nr) { return knn_layout(nq, nr).total; }

template <class T> static T* wat(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }

// query == nullptr / nq == 0: every reference point against the others
hipError_t knn_mean_dist2(int nq, const float* query, int nr, const float* reference, int K, int take_sqrt, float* out, void* ws,
                          size_t ws_bytes, int rank_mode, hipStream_t s) {
    const bool self = query == nullptr;
    if (self) nq = 0;
    const KnnLayout L = knn_layout(nq, nr);
    if (ws_bytes < L.total) return hipErrorInvalidValue;
    float* partial = wat<float>(ws, L.partial); float* bounds = wat<float>(ws, L.bounds);
    hipLaunchKernelGGL(knn_bounds_partial_kernel, dim3(L.blocks_r), dim3(kKnnThreads), 0, s, reference, nr, partial);
    if (!self) hipLaunchKernelGGL(knn_bounds_partial_kernel, dim3(L.blocks_q), dim3(kKnnThreads), 0, s, query, nq, partial + 6 * L.blocks_r);
    hipLaunchKernelGGL(knn_bounds_final_kernel, dim3(1), dim3(64), 0, s, partial, L.blocks_r + L.blocks_q, bounds);
    const size_t temp_bytes = radix_sort_temp_bytes((uint32_t)(nq > nr ? nq : nr));
    auto order_cloud = [&](const float* pts, int n, size_t codes, size_t codes_sorted, size_t order, size_t sorted) -> hipError_t {
        hipLaunchKernelGGL(knn_morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, bounds, wat<uint32_t>(ws, codes));
        hipError_t e = radix_sort_pairs(wat<uint32_t>(ws, codes), nullptr, wat<uint32_t>(ws, codes_sorted), wat<uint32_t>(ws, order),
                                        (uint32_t)n, 30, wat<void>(ws, L.sort_temp), temp_bytes, s, nullptr, nullptr, rank_mode);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(knn_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, wat<uint32_t>(ws, order), n, wat<float4>(ws, sorted));
        return hipGetLastError();
    };
    hipError_t e = order_cloud(reference, nr, L.r_codes, L.r_codes_sorted, L.r_order, L.r_sorted);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_boxes_kernel, dim3(L.n_boxes), dim3(kKnnThreads), 0, s, wat<float4>(ws, L.r_sorted), nr, wat<float4>(ws, L.boxes));
    if (!self) {
        e = order_cloud(query, nq, L.q_codes, L.q_codes_sorted, L.q_order, L.q_sorted);
        if (e != hipSuccess) return e;
    }
    const int n_search = self ? nr : nq;
    const dim3 grid((n_search + kKnnThreads - 1) / kKnnThreads), block(kKnnThreads);
    const float4* rs = wat<float4>(ws, L.r_sorted);
    const float4* qs = self ? rs : wat<float4>(ws, L.q_sorted);
    const uint32_t* rc = wat<uint32_t>(ws, L.r_codes_sorted);
    const uint32_t* qc = self ? rc : wat<uint32_t>(ws, L.q_codes_sorted);
    const float4* boxes = wat<float4>(ws, L.boxes);
#define SR_KNN_LAUNCH(KK, SELF) \
    hipLaunchKernelGGL((knn_search_kernel<KK, SELF>), grid, block, 0, s, qs, n_search, rs, rc, qc, nr, boxes, L.n_boxes, take_sqrt, out)
    if (K == 3) { if (self) SR_KNN_LAUNCH(3, true); else SR_KNN_LAUNCH(3, false); }
    else        { if (self) SR_KNN_LAUNCH(10, true); else SR_KNN_LAUNCH(10, false); }
#undef SR_KNN_LAUNCH
    return hipGetLastError();
}

}  // namespace sr
