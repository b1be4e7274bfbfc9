// What this box's HBM delivers to simple streaming kernels -- the ceiling next to the 8 TB/s vendor figure that the HBM-bound kernels
// (K1, K8, binning) are priced against ("record ... a measured hipMemcpyDtoD / triad ceiling").  Read-only, write-only, copy,
// triad with 16-B accesses per lane, grid-stride over 1 GiB arrays, plus hipMemcpyDtoD; best of 10 launches each, HIP events.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/hbm_ceiling.hip -o tools/ubench/hbm_ceiling && tools/ubench/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, size_t n, float* out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;   // (keeps the loads alive)
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n, float s) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i]; c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w); }
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b, *c; float* out;
    CK(hipMalloc((void**)&a, bytes)); CK(hipMalloc((void**)&b, bytes)); CK(hipMalloc((void**)&c, bytes)); CK(hipMalloc((void**)&out, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, %d MHz shader clock, %d MHz memory clock, %d-bit bus, %.0f GB\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000,
           p.memoryBusWidth, p.totalGlobalMem / 1e9);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int per_cu : {4, 8, 16, 32}) {
        const int grid = p.multiProcessorCount * per_cu;
        auto best = [&](auto launch, double moved) { float ms_best = 1e9f; for (int r = 0; r < 10; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms_best = std::min(ms_best, ms); } return moved / (ms_best * 1e-3) / 1e12; };
        const double r = best([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); }, (double)bytes);
        const double w = best([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n, 1.f); }, (double)bytes);
        const double cp = best([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        const double tr = best([&] { hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, 0, a, b, c, n, 0.5f); }, 3.0 * bytes);
        printf("%2d blocks of 256 per CU:  read %.2f TB/s   write %.2f   copy %.2f   triad %.2f   (1 GiB arrays, 16 B per lane, best of 10)\n", per_cu, r, w, cp, tr);
    }
    float ms_best = 1e9f;
    for (int r = 0; r < 10; ++r) { hipEventRecord(e0); hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms_best = std::min(ms_best, ms); }
    printf("hipMemcpyDtoD 1 GiB: %.2f TB/s (read + write)\n", 2.0 * bytes / (ms_best * 1e-3) / 1e12);
    return 0;
}
