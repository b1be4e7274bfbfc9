// Micro-benchmark: does a wave64 VALU instruction get cheaper when part of the wave is masked off in EXEC?  (If gfx950 skipped the
// inactive 32-lane half or 16-lane rows, clustering the contributing pixels of a quadrant test into one half would raise the blend
// kernels' effective lane utilisation for free.)  Measures ns per instruction and SIMD for v_fma_f32 / v_mul_f32 / v_exp_f32 with
// EXEC = all 64 lanes, the low 32, the low 16, one lane per row, and the odd lanes.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/exec_mask_ubench.hip -o /tmp/em && /tmp/em
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kUnroll = 8, CH = 8;
template <int OP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b, unsigned long long mask) {
    float x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    unsigned long long saved;
    int idx = blockIdx.x * 64 + threadIdx.x;   // every lane's output index is computed while all lanes are still enabled
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %2" : "=&s"(saved), "+v"(idx) : "s"(mask));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    asm volatile("s_mov_b64 exec, %2" : "+v"(idx), "+v"(s) : "s"(saved));   // nothing that all lanes need may float above this
    out[idx] = s;
}
template <int OP> void run(const char* name, float* d) {
    const int iters = 2000;
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x0001000100010001ull, 0xAAAAAAAAAAAAAAAAull, 0xFFFFFFFF00000000ull, 0xFFFF0000FFFF0000ull};
    const char* mn[] = {"all64", "low32", "low16", "1/row", "odd", "high32", "rows1,3"};
    for (int m = 0; m < 7; ++m) {
        printf("%-8s %-8s:", name, mn[m]);
        for (int w : {2, 4, 8}) {
            const int blocks = 256 * 4 * w;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f, masks[m]);
            hipEventRecord(e0); hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, masks[m]); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  w%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)w * iters * kUnroll * CH));
        }
        printf("\n");
    }
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<0>("fma", d); run<1>("mul", d); run<2>("exp", d);
    return 0;
}
