// render.hip -- K6 (per-tile front-to-back blend) and K7 (per-tile back-to-front backward).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; each wave owns an 8x8 pixel quadrant so a
// wave's footprint is compact (fewer splats overlap it -> more wave-uniform skips).  The tile's
// depth-sorted splat list is staged through LDS 256 records at a time (each thread gathers one
// 80-B packed record with five dwordx4 loads); in the inner loop all 64 lanes read the same LDS
// address (broadcast ds_read_b128, conflict-free).  No MFMA: there is no dense contraction here.
//
// K7 replaces the reference's ~18 global float atomics per (pixel, splat) pair with a wave-level
// multi-value transpose-reduction (24 cross-lane ops for 18 values) and ONE atomic per value per
// (wave, splat) -- and none at all when no lane of the wave is touched by the splat.
//
// Behavioural contract: SURVEY.md Appendix A.4 / A.5; output channel order
// [REF /root/reference/gaussian_renderer/__init__.py:149-165].
#include "common.h"

namespace sr {

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct Hit {
    float sx, sy, dx, dy, depth, G, alpha, pz_inv;
    float kx, ky, kz, lx, ly, lz;
    bool use3d;
};

// Ray-splat intersection + alpha for pixel (pxf, pyf). Returns false when the entry is skipped.
__device__ __forceinline__ bool intersect(float pxf, float pyf, const float4 q0, const float4 q1, const float4 q2, Hit& h) {
    const float Tux = q0.x, Tuy = q0.y, Tuz = q0.z, Tvx = q0.w, Tvy = q1.x, Tvz = q1.y, Twx = q1.z, Twy = q1.w, Twz = q2.x;
    h.kx = pxf * Twx - Tux; h.ky = pxf * Twy - Tuy; h.kz = pxf * Twz - Tuz;
    h.lx = pyf * Twx - Tvx; h.ly = pyf * Twy - Tvy; h.lz = pyf * Twz - Tvz;
    const float ppx = h.ky * h.lz - h.kz * h.ly;
    const float ppy = h.kz * h.lx - h.kx * h.lz;
    const float ppz = h.kx * h.ly - h.ky * h.lx;
    if (ppz == 0.f) return false;
    h.pz_inv = fast_rcp(ppz);
    h.sx = ppx * h.pz_inv; h.sy = ppy * h.pz_inv;
    const float rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = q2.y - pxf; h.dy = q2.z - pyf;
    const float rho2d = kFilterInvSquare * (h.dx * h.dx + h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = h.use3d ? (h.sx * Twx + h.sy * Twy) + Twz : Twz;
    if (h.depth < kNear) return false;
    const float power = -0.5f * rho;
    if (power > 0.f) return false;
    h.G = __expf(power);
    h.alpha = fminf(kAlphaCap, q2.w * h.G);
    if (h.alpha < kAlphaFloor) return false;
    return true;
}

__device__ __forceinline__ void pixel_of(int tile, int tiles_x, int tid, int& px, int& py) {
    const int wave = tid >> 6, lane = tid & 63;
    px = (tile % tiles_x) * kTile + (wave & 1) * 8 + (lane & 7);
    py = (tile / tiles_x) * kTile + (wave >> 1) * 8 + (lane >> 3);
}

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_forward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                                 const uint32_t* __restrict__ point_list,
                                                                 const float4* __restrict__ recs,
                                                                 float* __restrict__ out_color, float* __restrict__ out_allmap,
                                                                 float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ float4 s_q[kRecQuads][kBlock];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    int px, py;
    pixel_of(tile, f.tiles_x, tid, px, py);
    const bool inside = px < f.W && py < f.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dsum = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0xFFFFFFFFu;

    for (uint32_t base = range.x; base < range.y; base += kBlock) {
        if (__syncthreads_and(done)) break;  // also fences the previous round's LDS reads
        const uint32_t n = min((uint32_t)kBlock, range.y - base);
        if ((uint32_t)tid < n) {
            const float4* r = recs + (size_t)point_list[base + tid] * kRecQuads;
#pragma unroll
            for (int q = 0; q < kRecQuads; ++q) s_q[q][tid] = r[q];
        }
        __syncthreads();
        const uint32_t c0 = base - range.x;
        for (uint32_t j = 0; j < n; ++j) {
            if (__ballot(!done) == 0) break;  // whole wave finished
            Hit h;
            const float4 q2 = s_q[2][j];
            const bool valid = !done && intersect(pxf, pyf, s_q[0][j], s_q[1][j], q2, h);
            if (__ballot(valid) == 0) continue;  // splat misses this wave's 8x8 quadrant entirely
            if (valid) {
                const float test_T = T * (1.f - h.alpha);
                if (test_T < kTStop) {
                    done = true;  // this entry is NOT blended
                } else {
                    const float4 q3 = s_q[3][j], q4 = s_q[4][j];
                    const float w = h.alpha * T;
                    const float A = 1.f - T;
                    const float m = kFar / (kFar - kNear) * (1.f - kNear * fast_rcp(h.depth));
                    distortion += (m * m * A + M2 - 2.f * m * M1) * w;
                    Dsum += h.depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (T > 0.5f) { median_depth = h.depth; median_contributor = c0 + j + 1; }
                    N0 += q3.x * w; N1 += q3.y * w; N2 += q3.z * w;
                    C0 += q4.x * w; C1 += q4.y * w; C2 += q4.z * w;
                    T = test_T;
                    last_contributor = c0 + j + 1;
                }
            }
        }
    }
    if (inside) {
        const size_t HW = (size_t)f.H * f.W, pix = (size_t)py * f.W + px;
        final_T[pix] = T; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
        n_contrib[pix] = last_contributor; n_contrib[HW + pix] = median_contributor;
        out_color[pix] = C0 + T * f.bg[0];
        out_color[HW + pix] = C1 + T * f.bg[1];
        out_color[2 * HW + pix] = C2 + T * f.bg[2];
        out_allmap[pix] = Dsum;
        out_allmap[HW + pix] = 1.f - T;
        out_allmap[2 * HW + pix] = N0; out_allmap[3 * HW + pix] = N1; out_allmap[4 * HW + pix] = N2;
        out_allmap[5 * HW + pix] = median_depth;
        out_allmap[6 * HW + pix] = distortion;
    }
}

// ---------------------------------------------------------------------------------------------
// wave-level transpose-reduction: N values per lane -> lane l returns the 64-lane total of value
// index l / (64/N).  Stage with mask m folds the value set in half (upper-half lanes keep the upper
// half of the values), so the cost is N-1 cross-lane adds + log2(64/N) butterflies instead of 6*N.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float wave_reduce_multi(float (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two");
    int m = 32;
#pragma unroll
    for (int n = N; n > 1; n >>= 1, m >>= 1) {
        const bool hi = (lane & m) != 0;
#pragma unroll
        for (int k = 0; k < n / 2; ++k) {
            const float keep = hi ? v[k + n / 2] : v[k];
            const float send = hi ? v[k] : v[k + n / 2];
            v[k] = keep + __shfl_xor(send, m);
        }
    }
    float r = v[0];
#pragma unroll
    for (int mm = (64 / N) >> 1; mm > 0; mm >>= 1) r += __shfl_xor(r, mm);
    return r;
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_backward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const float4* __restrict__ recs,
                                                                  const float* __restrict__ final_T,
                                                                  const uint32_t* __restrict__ n_contrib,
                                                                  const float* __restrict__ dL_dcolor,
                                                                  const float* __restrict__ dL_dallmap,
                                                                  float* __restrict__ grecs) {
    __shared__ float4 s_q[kRecQuads][kBlock];
    __shared__ uint32_t s_gid[kBlock];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.x;
    int px, py;
    pixel_of(tile, f.tiles_x, tid, px, py);
    const bool inside = px < f.W && py < f.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)f.H * f.W, pix = inside ? (size_t)py * f.W + px : 0;

    const float T_final = inside ? final_T[pix] : 0.f;
    const float final_D = inside ? final_T[HW + pix] : 0.f;
    const float final_D2 = inside ? final_T[2 * HW + pix] : 0.f;
    const float final_A = 1.f - T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    const uint32_t median_contributor = inside ? n_contrib[HW + pix] : 0u;
    float gpix[3] = {0, 0, 0}, gN[3] = {0, 0, 0}, g_depth = 0.f, g_accum = 0.f, g_median = 0.f, g_reg = 0.f;
    if (inside) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { gpix[c] = dL_dcolor[c * HW + pix]; gN[c] = dL_dallmap[(2 + c) * HW + pix]; }
        g_depth = dL_dallmap[pix]; g_accum = dL_dallmap[HW + pix];
        g_median = dL_dallmap[5 * HW + pix]; g_reg = dL_dallmap[6 * HW + pix];
    }
    const float bg_dot = f.bg[0] * gpix[0] + f.bg[1] * gpix[1] + f.bg[2] * gpix[2];

    if (tid == 0) s_max = 0;
    __syncthreads();
    // deepest entry any pixel of this wave / this tile needs
    uint32_t wave_last = last_contributor;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, m));
    if (lane == 0) atomicMax(&s_max, wave_last);
    __syncthreads();
    const uint32_t total = s_max;

    float T = T_final;
    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.f;
    float last_depth = 0.f, accum_depth_rec = 0.f, accum_alpha_rec = 0.f;
    float last_normal[3] = {0, 0, 0}, accum_normal_rec[3] = {0, 0, 0}, last_dL_dT = 0.f;

    const int rounds = (int)((total + kBlock - 1) / kBlock);
    for (int rd = rounds - 1; rd >= 0; --rd) {
        __syncthreads();
        const uint32_t rbase = (uint32_t)rd * kBlock;
        const uint32_t n = min((uint32_t)kBlock, total - rbase);
        if ((uint32_t)tid < n) {
            const uint32_t gid = point_list[range.x + rbase + tid];
            s_gid[tid] = gid;
            const float4* r = recs + (size_t)gid * kRecQuads;
#pragma unroll
            for (int q = 0; q < kRecQuads; ++q) s_q[q][tid] = r[q];
        }
        __syncthreads();
        if (wave_last <= rbase) continue;  // nothing in this round for this wave
        const int jstart = (int)min(n, wave_last - rbase) - 1;
        for (int j = jstart; j >= 0; --j) {
            const uint32_t cidx = rbase + (uint32_t)j;  // 0-based contributor index
            Hit h;
            const float4 q0 = s_q[0][j], q1 = s_q[1][j], q2 = s_q[2][j];
            const bool valid = (cidx < last_contributor) && intersect(pxf, pyf, q0, q1, q2, h);
            if (__ballot(valid) == 0) continue;
            float vA[16], vB[4];
#pragma unroll
            for (int k = 0; k < 16; ++k) vA[k] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) vB[k] = 0.f;
            if (valid) {
                const float4 q3 = s_q[3][j], q4 = s_q[4][j];
                const float Twx = q1.z, Twy = q1.w;
                const float one_m_inv = fast_rcp(1.f - h.alpha);
                T = T * one_m_inv;
                const float w = h.alpha * T;
                float dL_dalpha = 0.f;
                const float col[3] = {q4.x, q4.y, q4.z}, nrm[3] = {q3.x, q3.y, q3.z};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    accum_rec[c] = last_alpha * last_color[c] + (1.f - last_alpha) * accum_rec[c];
                    last_color[c] = col[c];
                    dL_dalpha += (col[c] - accum_rec[c]) * gpix[c];
                    vB[c] = w * gpix[c];
                }
                float dL_dz = 0.f;
                const float inv_depth = fast_rcp(h.depth);
                const float m_d = kFar / (kFar - kNear) * (1.f - kNear * inv_depth);
                const float dmd_dd = (kFar * kNear) / (kFar - kNear) * inv_depth * inv_depth;
                if (cidx == median_contributor - 1u) dL_dz += g_median;
                const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * g_reg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * h.alpha + (1.f - h.alpha) * last_dL_dT;
                const float dL_dmd = 2.f * w * (m_d * final_A - final_D) * g_reg;
                dL_dz += dL_dmd * dmd_dd;
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = h.depth;
                dL_dalpha += (h.depth - accum_depth_rec) * g_depth;
                accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                dL_dalpha += (1.f - accum_alpha_rec) * g_accum;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    accum_normal_rec[c] = last_alpha * last_normal[c] + (1.f - last_alpha) * accum_normal_rec[c];
                    last_normal[c] = nrm[c];
                    dL_dalpha += (nrm[c] - accum_normal_rec[c]) * gN[c];
                    vA[12 + c] = w * gN[c];
                }
                dL_dalpha *= T;
                last_alpha = h.alpha;
                dL_dalpha += (-T_final * one_m_inv) * bg_dot;
                const float dL_dG = q2.w * dL_dalpha;
                dL_dz += w * g_depth;
                if (h.use3d) {
                    const float dLdsx = dL_dG * -h.G * h.sx + dL_dz * Twx;
                    const float dLdsy = dL_dG * -h.G * h.sy + dL_dz * Twy;
                    const float ax = dLdsx * h.pz_inv, ay = dLdsy * h.pz_inv;
                    const float dpx = ax, dpy = ay, dpz = -(ax * h.sx + ay * h.sy);
                    // dL_dk = l x dp ; dL_dl = dp x k
                    const float dkx = h.ly * dpz - h.lz * dpy, dky = h.lz * dpx - h.lx * dpz, dkz = h.lx * dpy - h.ly * dpx;
                    const float dlx = dpy * h.kz - dpz * h.ky, dly = dpz * h.kx - dpx * h.kz, dlz = dpx * h.ky - dpy * h.kx;
                    vA[0] = -dkx; vA[1] = -dky; vA[2] = -dkz;
                    vA[3] = -dlx; vA[4] = -dly; vA[5] = -dlz;
                    vA[6] = pxf * dkx + pyf * dlx + dL_dz * h.sx;
                    vA[7] = pxf * dky + pyf * dly + dL_dz * h.sy;
                    vA[8] = pxf * dkz + pyf * dlz + dL_dz;
                } else {
                    vA[9] = dL_dG * (-h.G * kFilterInvSquare * h.dx);
                    vA[10] = dL_dG * (-h.G * kFilterInvSquare * h.dy);
                    vA[8] = dL_dz;
                }
                vA[11] = h.G * dL_dalpha;
            }
            const float rA = wave_reduce_multi<16>(vA, lane);
            const float rB = wave_reduce_multi<4>(vB, lane);
            float* g = grecs + (size_t)s_gid[j] * kRecFloats;
            if ((lane & 3) == 0 && (lane >> 2) < 15) atomicAdd(g + (lane >> 2), rA);
            if ((lane & 15) == 0 && (lane >> 4) < 3) atomicAdd(g + 16 + (lane >> 4), rB);
        }
    }
}

// launchers ---------------------------------------------------------------------------------------
hipError_t launch_render_forward(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                 float* out_color, float* out_allmap, float* final_T, uint32_t* n_contrib, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(render_forward_kernel, dim3(n_tiles), dim3(kBlock), 0, s, f, ranges, point_list, recs, out_color,
                       out_allmap, final_T, n_contrib);
    return hipGetLastError();
}

hipError_t launch_render_backward(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                  const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                  const float* dL_dallmap, float* grecs, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(render_backward_kernel, dim3(n_tiles), dim3(kBlock), 0, s, f, ranges, point_list, recs, final_T,
                       n_contrib, dL_dcolor, dL_dallmap, grecs);
    return hipGetLastError();
}

}  // namespace sr
