// Does ds_add_rtn_u32 hand out its return values in ascending lane order when several lanes of ONE wave instruction hit the same
// LDS address?  (Undocumented; a stable ranking by LDS atomics would replace the match-any ballots of the radix / expanding passes.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic_order.hip -o /tmp/lo && /tmp/lo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__global__ __launch_bounds__(256) void k(const uint32_t* digits, uint32_t* ranks, int bins) {
    __shared__ uint32_t s_cnt[4][1024];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 4 * 1024; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    for (int it = 0; it < 8; ++it) {
        const uint32_t d = digits[(blockIdx.x * 8 + it) * 256 + tid] % bins;
        ranks[(blockIdx.x * 8 + it) * 256 + tid] = atomicAdd(&s_cnt[w][d], 1u);
    }
}
int main() {
    const int blocks = 4096, n = blocks * 8 * 256;
    std::vector<uint32_t> h(n), r(n);
    std::mt19937 rng(1);
    uint32_t *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    long bad_total = 0;
    for (int bins : {1, 2, 3, 7, 16, 68, 120, 256, 1000}) {
        for (auto& x : h) x = rng();
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, o, bins);
        hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < 4; ++w) {
                std::vector<uint32_t> cnt(bins, 0);
                for (int it = 0; it < 8; ++it)
                    for (int l = 0; l < 64; ++l) {
                        const int idx = (b * 8 + it) * 256 + w * 64 + l;
                        const uint32_t dg = h[idx] % bins;
                        if (r[idx] != cnt[dg]) ++bad;
                        ++cnt[dg];
                    }
            }
        printf("bins %4d: %ld of %d returns out of lane order\n", bins, bad, n);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
