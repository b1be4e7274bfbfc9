// Micro-benchmark: how many wave64 VALU instructions per cycle ONE SIMD sustains as a function of resident waves,
// in shader cycles (s_memtime), for independent and dependent streams.  Decides whether the blend kernels are at the VALU roof.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue_ubench.hip -o /tmp/vi && /tmp/vi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kUnroll = 8;
template <int OP, int CH>   // CH independent chains per wave
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
    float x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
                else if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                else if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                else if (OP == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
                else if (OP == 6) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                else if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == 9) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(x[i]) : "v"(a), "v"(b) : "s20", "s21");
                else if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[i]) : "v"(a));
                else if (OP == 11) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(x[i]));
                else if (OP == 12) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i]), __float_as_uint(x[(i + 1) % CH]), false, false); x[i] = __uint_as_float(r[0]); x[(i + 1) % CH] = __uint_as_float(r[1]); }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void kpk(float* out, int iters, float a, float b) {
    float2v x[8], aa = {a, a}, bb = {b, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = float2v{a + threadIdx.x * 1e-3f + i, b + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(aa), "v"(bb));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void kmov64(float* out, int iters, float a, float b) {
    float2v x[8], bb = {b, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = float2v{a + threadIdx.x * 1e-3f + i, b + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64 %0, %1" : "+v"(x[i]) : "v"(bb));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP, int CH> void run(const char* name, float* d, unsigned long long* c, int per_op) {
    const int iters = 2000;
    printf("%-22s chains=%d:", name, CH);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * 4 * w;
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, d, c, iters, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
        // s_memtime counts at a fixed 100 MHz-ish?  report both raw ticks/instr and wall-based rate
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, d, c, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)w * iters * kUnroll * CH * per_op;
        printf("  w%d %.2f ns/instr/SIMD (ticks/instr %.2f)", w, ms * 1e6 / instr_per_simd, mean * w / instr_per_simd);
    }
    printf("\n");
}
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 4 * 8 * 64 * 4); hipMalloc(&c, 256 * 4 * 8 * 8);
    run<0, 8>("fma vvv", d, c, 1); run<0, 1>("fma vvv", d, c, 1); run<0, 2>("fma vvv", d, c, 1);
    run<1, 8>("fma svv", d, c, 1);
    run<2, 8>("mul", d, c, 1);
    run<3, 8>("exp", d, c, 1); run<4, 8>("rcp", d, c, 1);
    run<5, 8>("cmp+cndmask(2)", d, c, 2);
    run<6, 8>("add dpp row_ror", d, c, 1); run<6, 1>("add dpp row_ror", d, c, 1);
    run<7, 8>("mov", d, c, 1);
    run<8, 8>("fmac vv (VOP2)", d, c, 1);
    run<9, 8>("cmp_e64->sgpr+cndmask_e64(2)", d, c, 2);
    run<10, 8>("fma with inline const", d, c, 1);
    run<11, 8>("mul with literal (VOP2)", d, c, 1);
    run<12, 8>("permlane32_swap", d, c, 1);
    {   // packed FP32: two FMAs per instruction
        printf("pk_fma (2 FMAs/instr) chains=8:");
        for (int w : {1, 2, 4, 8}) {
            const int blocks = 256 * 4 * w, iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kpk, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
            hipEventRecord(e0); hipLaunchKernelGGL(kpk, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  w%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)w * iters * kUnroll * 8));
        }
        printf("\n");
    }
    {   // 64-bit move: two registers per instruction
        printf("mov_b64 (2 regs/instr) chains=8:");
        for (int w : {1, 2, 3, 4, 8}) {
            const int blocks = 256 * 4 * w, iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kmov64, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
            hipEventRecord(e0); hipLaunchKernelGGL(kmov64, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  w%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)w * iters * kUnroll * 8));
        }
        printf("\n");
    }
    return 0;
}
