// Micro-benchmark: cost of the 24-value wave transpose-reduction variants on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/reduce_ubench.hip -o /tmp/reduce_ubench && /tmp/reduce_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void fold32(float& a, float& b) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
__device__ __forceinline__ void fold16(float& a, float& b) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
template <int C> __device__ __forceinline__ float dppm(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xf, 0xf, false)); }
__device__ __forceinline__ float row_sum16(float x) { x += dppm<0x128>(x); x += dppm<0x124>(x); x += dppm<0x122>(x); x += dppm<0x121>(x); return x; }

// in-row (16-lane) fold of a pair: lanes of the upper set keep b, the others a; partner via a DPP permutation
template <int C> __device__ __forceinline__ float fold_dpp(bool hi, float a, float b) {
    const float keep = hi ? b : a, send = hi ? a : b;
    return keep + dppm<C>(send);
}
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    __shared__ float s_acc[64][24];
    const int lane = threadIdx.x;
    if (MODE == 9) { for (int i = 0; i < 24; ++i) s_acc[lane][i] = 0.f; }
    float v[24]; float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = (float)(lane * 24 + i + it) * 1e-3f + acc * 1e-9f;
        if (MODE == 1) {   // swap-fold + dpp row sums (the K7 reduction)
#pragma unroll
            for (int i = 0; i < 12; ++i) fold32(v[i], v[i + 12]);
#pragma unroll
            for (int i = 0; i < 6; ++i) fold16(v[i], v[i + 6]);
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = row_sum16(v[i]);
        } else if (MODE == 2) {  // plain butterfly with __shfl_xor
#pragma unroll
            for (int i = 0; i < 24; ++i) for (int m = 32; m > 0; m >>= 1) v[i] += __shfl_xor(v[i], m);
        } else if (MODE == 3) {  // only the permlane folds
#pragma unroll
            for (int i = 0; i < 12; ++i) fold32(v[i], v[i + 12]);
#pragma unroll
            for (int i = 0; i < 6; ++i) fold16(v[i], v[i + 6]);
        } else if (MODE == 6) {  // swap folds for 32/16, then select+DPP folds inside the row
#pragma unroll
            for (int i = 0; i < 12; ++i) fold32(v[i], v[i + 12]);
#pragma unroll
            for (int i = 0; i < 6; ++i) fold16(v[i], v[i + 6]);
            // 6 values per row -> fold by 8: 3, by 4: (pad) 2, by 2: 1, xor 1
            const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { const float keep = h8 ? v[i + 3] : v[i], send = h8 ? v[i] : v[i + 3]; v[i] = keep + dppm<0x128>(send); }
            { const float keep = h4 ? v[2] : v[0], send = h4 ? v[0] : v[2]; const float k2 = h4 ? 0.f : v[1], s2 = h4 ? v[1] : 0.f;
              v[0] = keep + __shfl_xor(send, 4); v[1] = k2 + __shfl_xor(s2, 4); }
            { const float keep = h2 ? v[1] : v[0], send = h2 ? v[0] : v[1]; v[0] = keep + __shfl_xor(send, 2); }
            v[0] += __shfl_xor(v[0], 1);
            v[1] = v[2] = v[3] = v[4] = v[5] = 0.f;
        } else if (MODE == 7) {  // select + ds_bpermute folds for 32/16, then in-row folds
            const bool h32 = (lane & 32) != 0, h16 = (lane & 16) != 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float keep = h32 ? v[i + 12] : v[i], send = h32 ? v[i] : v[i + 12]; v[i] = keep + __shfl_xor(send, 32); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { const float keep = h16 ? v[i + 6] : v[i], send = h16 ? v[i] : v[i + 6]; v[i] = keep + __shfl_xor(send, 16); }
            const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) { const float keep = h8 ? v[i + 3] : v[i], send = h8 ? v[i] : v[i + 3]; v[i] = keep + dppm<0x128>(send); }
            { const float keep = h4 ? v[2] : v[0], send = h4 ? v[0] : v[2]; const float k2 = h4 ? 0.f : v[1], s2 = h4 ? v[1] : 0.f;
              v[0] = keep + __shfl_xor(send, 4); v[1] = k2 + __shfl_xor(s2, 4); }
            { const float keep = h2 ? v[1] : v[0], send = h2 ? v[0] : v[1]; v[0] = keep + __shfl_xor(send, 2); }
            v[0] += __shfl_xor(v[0], 1);
            v[1] = v[2] = v[3] = v[4] = v[5] = 0.f;
        } else if (MODE == 8) {  // like 7 but row_sum16 for the in-row part (6 values x 4 dpp)
            const bool h32 = (lane & 32) != 0, h16 = (lane & 16) != 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float keep = h32 ? v[i + 12] : v[i], send = h32 ? v[i] : v[i + 12]; v[i] = keep + __shfl_xor(send, 32); }
#pragma unroll
            for (int i = 0; i < 6; ++i) { const float keep = h16 ? v[i + 6] : v[i], send = h16 ? v[i] : v[i + 6]; v[i] = keep + __shfl_xor(send, 16); }
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = row_sum16(v[i]);
        } else if (MODE == 9) {  // per-row folds (mirror DPPs) 24->12->6->3->2, then 4 rows combine with LDS float atomics
            const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0, h1 = (lane & 1) != 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = fold_dpp<0x140>(h8, v[i], v[i + 12]);   // row_mirror
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = fold_dpp<0x141>(h4, v[i], v[i + 6]);     // row_half_mirror
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = fold_dpp<0x1B>(h2, v[i], v[i + 3]);      // quad_perm [3,2,1,0]
            const float r0 = fold_dpp<0xB1>(h1, v[0], v[2]);                              // quad_perm [1,0,3,2]
            const float r1 = fold_dpp<0xB1>(h1, v[1], 0.f);
            const int base = (h8 ? 12 : 0) + (h4 ? 6 : 0) + (h2 ? 3 : 0);
            float* o = &s_acc[it & 63][base + (h1 ? 2 : 0)];
            atomicAdd(o, r0);
            if (!h1) atomicAdd(o + 1, r1);
            v[0] = r0; v[1] = r1; v[2] = v[3] = v[4] = v[5] = 0.f;
        } else if (MODE == 4) {  // only dpp row sums of 6 values
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = row_sum16(v[i]);
        }
        acc += v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
    }
    if (MODE == 9) acc += s_acc[lane][lane % 24];
    out[blockIdx.x * 64 + lane] = acc;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * 4;  // 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 16 * 64 * 4);
    const int iters = 2000;
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters), t5 = 0, t6 = run<6>(d, iters), t7 = run<7>(d, iters), t8 = run<8>(d, iters), t9 = run<9>(d, iters);
    // per SIMD: 4 waves x iters reductions; cycles per reduction per SIMD-slot at ~2.1 GHz
    auto cyc = [&](float ms) { return (ms - t0) * 1e-3 * 2.1e9 / (4.0 * iters); };
    printf("baseline %.3f ms | swap+dpp %.3f ms (%.0f cyc/reduce) | shfl butterfly %.3f ms (%.0f) | folds only %.3f (%.0f) | dpp rows only %.3f (%.0f)\n",
           t0, t1, cyc(t1), t2, cyc(t2), t3, cyc(t3), t4, cyc(t4));
    printf("swap folds + in-row folds %.3f ms (%.0f) | bpermute folds + in-row folds %.3f ms (%.0f) | bpermute folds + dpp rows %.3f (%.0f)\n", t6, cyc(t6), t7, cyc(t7), t8, cyc(t8));
    printf("per-row mirror folds + LDS atomics %.3f ms (%.0f cyc/reduce)\n", t9, cyc(t9));
    return 0;
}
