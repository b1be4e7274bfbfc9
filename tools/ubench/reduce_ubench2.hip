// Micro-benchmark 2: swap-based vs LDS-based folds for the 24-value wave reduction (opaque inputs from memory).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void fold32(float& a, float& b) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
__device__ __forceinline__ void fold16(float& a, float& b) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
template <int C> __device__ __forceinline__ float dppm(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xf, 0xf, false)); }
__device__ __forceinline__ float inrow(float (&v)[24], int lane) {
    const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float keep = h8 ? v[k + 3] : v[k], send = h8 ? v[k] : v[k + 3]; v[k] = keep + dppm<0x128>(send); }
    { const float k0 = h4 ? v[2] : v[0], s0 = h4 ? v[0] : v[2], k1 = h4 ? 0.f : v[1], s1 = h4 ? v[1] : 0.f; v[0] = k0 + __shfl_xor(s0, 4); v[1] = k1 + __shfl_xor(s1, 4); }
    { const float k0 = h2 ? v[1] : v[0], s0 = h2 ? v[0] : v[1]; v[0] = k0 + __shfl_xor(s0, 2); }
    return v[0] + __shfl_xor(v[0], 1);
}
constexpr int kStride = 28;  // floats per lane slot (112 B): conflict-free b128 access
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* __restrict__ in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s_red[64 * kStride];
    const int lane = threadIdx.x;
    float v[24]; float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = in[(it & 15) * 64 * 24 + i * 64 + lane] + acc * 1e-9f;
        float r = 0.f;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) fold32(v[i], v[i + 12]);
#pragma unroll
            for (int i = 0; i < 6; ++i) fold16(v[i], v[i + 6]);
            r = inrow(v, lane);
        } else if (MODE == 2) {
            float4* mine = reinterpret_cast<float4*>(s_red + lane * kStride);
#pragma unroll
            for (int i = 0; i < 6; ++i) mine[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            {
                const int off = (lane & 32) ? 3 : 0;
                const float4* own = reinterpret_cast<const float4*>(s_red + lane * kStride) + off;
                const float4* oth = reinterpret_cast<const float4*>(s_red + (lane ^ 32) * kStride) + off;
#pragma unroll
                for (int i = 0; i < 3; ++i) { const float4 a = own[i], b = oth[i]; v[4 * i] = a.x + b.x; v[4 * i + 1] = a.y + b.y; v[4 * i + 2] = a.z + b.z; v[4 * i + 3] = a.w + b.w; }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) mine[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            {
                const int off = (lane & 16) ? 6 : 0;   // floats
                const float2* own = reinterpret_cast<const float2*>(s_red + lane * kStride + off);
                const float2* oth = reinterpret_cast<const float2*>(s_red + (lane ^ 16) * kStride + off);
#pragma unroll
                for (int i = 0; i < 3; ++i) { const float2 a = own[i], b = oth[i]; v[2 * i] = a.x + b.x; v[2 * i + 1] = a.y + b.y; }
            }
            r = inrow(v, lane);
        } else if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) r += v[i];
        }
        acc += r;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
template <int MODE> float run(const float* in, float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * 2;  // 2 waves per SIMD, like K7
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, in, d, 10);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, in, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *d, *in; hipMalloc(&d, 256 * 16 * 64 * 4); hipMalloc(&in, 16 * 64 * 24 * 4);
    hipMemset(in, 0, 16 * 64 * 24 * 4);
    const int iters = 2000;
    float t0 = run<0>(in, d, iters), t1 = run<1>(in, d, iters), t2 = run<2>(in, d, iters);
    auto cyc = [&](float ms) { return (ms - t0) * 1e-3 * 2.1e9 / (2.0 * iters); };
    printf("baseline %.3f ms | swap folds + in-row %.3f ms (%.0f cyc) | LDS folds + in-row %.3f ms (%.0f cyc)\n", t0, t1, cyc(t1), t2, cyc(t2));
    return 0;
}
