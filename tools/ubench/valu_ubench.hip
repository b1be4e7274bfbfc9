// Micro-benchmark: SIMD issue cost (cycles per wave64 instruction) of the VALU ops the blend kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kChains = 8, kUnroll = 16;
template <int OP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    float2 p[kChains / 2];
    for (int i = 0; i < kChains / 2; ++i) p[i] = make_float2(x[2 * i], x[2 * i + 1]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int i = 0; i < kChains; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
                else if (OP == 1) x[i] = x[i] * a;
                else if (OP == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
                else if (OP == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
                else if (OP == 4) x[i] = x[i] > b ? x[i] * a : x[i];          // cmp + cndmask/mul mix
                else if (OP == 5) x[i] = fminf(x[i], b) + a;
                else if (OP == 7) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x128, 0xf, 0xf, false)) + a;
            }
            if (OP == 6) {
#pragma unroll
                for (int i = 0; i < kChains / 2; ++i) {   // packed fma: 2 results per instruction
                    float2 t = p[i];
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
                    p[i] = t;
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < kChains; ++i) s += x[i];
    for (int i = 0; i < kChains / 2; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP> double run(float* d, int iters, int waves_per_simd, int ops_per_iter) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.1e9 / ((double)waves_per_simd * iters * ops_per_iter);   // cycles (at 2.1 GHz) per wave-instruction per SIMD
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    const int it = 2000, n = kChains * kUnroll;
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d  cycles per wave-instr: fma %.2f | mul %.2f | rcp %.2f | exp2 %.2f | cmp+sel+mul (3 ops) %.2f | min+add (2 ops) %.2f | pk_fma %.2f | dpp mov+add (2 ops) %.2f\n",
               w, run<0>(d, it, w, n), run<1>(d, it, w, n), run<2>(d, it, w, n), run<3>(d, it, w, n), run<4>(d, it, w, n), run<5>(d, it, w, n),
               run<6>(d, it, w, n / 2), run<7>(d, it, w, n));
    }
    return 0;
}
