// render.hip -- K6 (per-tile front-to-back blend) and K7 (per-tile back-to-front backward).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; each wave owns an 8x8 pixel quadrant so a
// wave's footprint is compact (fewer splats overlap it -> more wave-uniform skips).  The tile's
// depth-sorted splat list is staged through LDS 256 records at a time (each thread gathers one
// 80-B packed record with five dwordx4 loads); in the inner loop all 64 lanes read the same LDS
// address (broadcast ds_read_b128, conflict-free).  No MFMA: there is no dense contraction here.
//
// K7 replaces the reference's ~18 global float atomics per (pixel, splat) pair with a wave-level
// multi-value transpose-reduction (v_permlane32/16_swap + DPP), an LDS combine across the tile's 4
// waves and ONE plain 80-B store per (tile, splat): no global atomics, and nothing at all for a wave
// no lane of which is touched by the splat.
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
// Quadrant culling.  A list entry can only contribute to a pixel if alpha = min(0.99, opacity*G) >= 1/255,
// i.e. rho = min(rho3d, rho2d) <= thr = 2 ln(255 opacity).  {rho3d <= thr} is the image of the disc
// u^2+v^2 <= thr under the splat's homography -- an ellipse with dual conic C* = Q diag(thr,thr,-1) Q^T --
// and {rho2d <= thr} is a disc of radius sqrt(thr/2) around means2D.  Each staging thread bounds that
// union by an octagon (support in directions x, y, x+y, x-y from the tangent-line equation
// l^T C* l = 0) and tests it against the tile's four 8x8 quadrants.  Entries dropped here are entries the
// per-pixel test would skip anyway (`continue` in Appendix A.4), so results are unchanged; the bound has
// 0.3 px / 1 % slack for float rounding and keeps the entry whenever anything is degenerate or NaN.
// Coordinates are taken relative to the tile centre to avoid cancellation in the conic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quadrant_mask(const float4 q0, const float4 q1, const float4 q2, float Xc, float Yc) {
    float thr = 2.f * __logf(255.f * q2.w);
    thr = thr * 1.01f + 0.01f;
    if (thr <= 0.f) return 0u;
    const float Tw0 = q1.z, Tw1 = q1.w, Tw2 = q2.x;
    const float Tu0 = q0.x - Xc * Tw0, Tu1 = q0.y - Xc * Tw1, Tu2 = q0.z - Xc * Tw2;
    const float Tv0 = q0.w - Yc * Tw0, Tv1 = q1.x - Yc * Tw1, Tv2 = q1.y - Yc * Tw2;
    const float c22 = thr * (Tw0 * Tw0 + Tw1 * Tw1) - Tw2 * Tw2;
    if (!(c22 < 0.f)) return 0xFu;  // the cutoff disc reaches the camera plane: unbounded footprint
    const float c00 = thr * (Tu0 * Tu0 + Tu1 * Tu1) - Tu2 * Tu2;
    const float c01 = thr * (Tu0 * Tv0 + Tu1 * Tv1) - Tu2 * Tv2;
    const float c11 = thr * (Tv0 * Tv0 + Tv1 * Tv1) - Tv2 * Tv2;
    const float c02 = thr * (Tu0 * Tw0 + Tu1 * Tw1) - Tu2 * Tw2;
    const float c12 = thr * (Tv0 * Tw0 + Tv1 * Tw1) - Tv2 * Tw2;
    const float inv = 1.f / c22;
    const float r = sqrtf(0.5f * thr);
    const float mx = q2.y - Xc, my = q2.z - Yc;
    float lo[4], hi[4];
    const float A[4] = {c00, c11, c00 + 2.f * c01 + c11, c00 - 2.f * c01 + c11};
    const float B[4] = {c02, c12, c02 + c12, c02 - c12};
    const float ctr[4] = {mx, my, mx + my, mx - my};
    const float rad[4] = {r, r, r * 1.4142137f, r * 1.4142137f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float disc = B[d] * B[d] - A[d] * c22;
        const float half = sqrtf(fmaxf(disc, 0.f)) * (-inv) * 1.01f;
        const float dc = B[d] * inv;
        lo[d] = fminf(dc - half, ctr[d] - rad[d]);
        hi[d] = fmaxf(dc + half, ctr[d] + rad[d]);
    }
    const float m = 0.3f;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = (q & 1) ? 0.f : -8.f, x1 = x0 + 7.f;
        const float y0 = (q & 2) ? 0.f : -8.f, y1 = y0 + 7.f;
        const bool out = lo[0] > x1 + m || hi[0] < x0 - m || lo[1] > y1 + m || hi[1] < y0 - m ||
                         lo[2] > x1 + y1 + 2.f * m || hi[2] < x0 + y0 - 2.f * m ||
                         lo[3] > x1 - y0 + 2.f * m || hi[3] < x0 - y1 - 2.f * m;
        if (!out) mask |= 1u << q;
    }
    return mask;
}

// Wave `wave` compacts the indices (ascending) of the staged entries [0, n) whose mask has bit `wave` set and
// whose index is < limit, into s_list[wave][...]; returns the count.  Wave-local: no workgroup barrier needed.
__device__ __forceinline__ int build_wave_list(const uint8_t* s_mask, uint16_t (*s_list)[kBlock], int wave, int lane,
                                               uint32_t n, uint32_t limit) {
    int count = 0;
#pragma unroll
    for (int c = 0; c < kBlock / 64; ++c) {
        const uint32_t j = (uint32_t)(c * 64 + lane);
        const bool keep = j < n && j < limit && ((s_mask[j] >> wave) & 1u);
        const unsigned long long b = __ballot(keep);
        if (keep) s_list[wave][count + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)j;
        count += __popcll(b);
    }
    __builtin_amdgcn_wave_barrier();
    return count;
}

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_forward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                                 const uint32_t* __restrict__ point_list,
                                                                 const float4* __restrict__ recs,
                                                                 float* __restrict__ out_color, float* __restrict__ out_allmap,
                                                                 float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                                 int cull) {
    __shared__ float4 s_q[kRecQuads][kBlock];
    __shared__ uint8_t s_mask[kBlock];
    __shared__ uint16_t s_list[kBlock / 64][kBlock];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x;
    const float Xc = (float)((tile % f.tiles_x) * kTile + 8), Yc = (float)((tile / f.tiles_x) * kTile + 8);
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
            float4 q[kRecQuads];
#pragma unroll
            for (int k = 0; k < kRecQuads; ++k) { q[k] = r[k]; s_q[k][tid] = q[k]; }
            s_mask[tid] = cull ? (uint8_t)quadrant_mask(q[0], q[1], q[2], Xc, Yc) : (uint8_t)0xF;
        }
        __syncthreads();
        const uint32_t c0 = base - range.x;
        if (__ballot(!done) == 0) continue;  // this wave is finished (it still takes part in the barriers)
        const int cnt = build_wave_list(s_mask, s_list, wave, lane, n, n);
        for (int idx = 0; idx < cnt; ++idx) {
            const uint32_t j = s_list[wave][idx];
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
// wave-level transpose-reduction of 20 values per lane (gfx950):
//   fold across the two 32-lane halves with v_permlane32_swap (value k <-> k+10), across row pairs with
//   v_permlane16_swap (k <-> k+5), then a 4-step DPP row rotation sum.  50 VALU ops for 20 values (a plain
//   butterfly needs 120 cross-lane ops).  Afterwards every lane of 16-lane row g holds, in v[0..4], the
//   64-lane totals of values 5g .. 5g+4.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fold32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void fold16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int kCtrl>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0x128>(x);  // row_ror:8
    x += dpp_mov<0x124>(x);  // row_ror:4
    x += dpp_mov<0x122>(x);  // row_ror:2
    x += dpp_mov<0x121>(x);  // row_ror:1
    return x;
}
__device__ __forceinline__ void wave_reduce20(float (&v)[20]) {
#pragma unroll
    for (int k = 0; k < 10; ++k) fold32(v[k], v[k + 10]);
#pragma unroll
    for (int k = 0; k < 5; ++k) fold16(v[k], v[k + 5]);
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = row_sum16(v[k]);
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
// Output: one 80-B gradient record per (tile, Gaussian) duplicate, stored at the duplicate's EMISSION index
// (inst_grads[perm[pos]]), where the records of one Gaussian are contiguous; K8 sums them.  No global atomics: per (wave, splat) the 18 partial
// sums are wave-reduced, the 4 waves of the tile combine in LDS, and each record is stored exactly once
// (coalesced, 5 x dwordx4 per thread).  Records of list entries no pixel reached are written as zeros.
__global__ __launch_bounds__(kBlock) void render_backward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ point_list,
                                                                  const float4* __restrict__ recs,
                                                                  const float* __restrict__ final_T,
                                                                  const uint32_t* __restrict__ n_contrib,
                                                                  const float* __restrict__ dL_dcolor,
                                                                  const float* __restrict__ dL_dallmap,
                                                                  const uint32_t* __restrict__ perm,
                                                                  float4* __restrict__ inst_grads, int cull) {
    __shared__ float4 s_q[kRecQuads][kBlock];
    __shared__ __attribute__((aligned(16))) float s_acc[kBlock][kRecFloats];
    __shared__ uint8_t s_mask[kBlock];
    __shared__ uint16_t s_list[kBlock / 64][kBlock];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const float Xc = (float)((tile % f.tiles_x) * kTile + 8), Yc = (float)((tile / f.tiles_x) * kTile + 8);
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
    {
        float4* z = reinterpret_cast<float4*>(&s_acc[tid][0]);
#pragma unroll
        for (int q = 0; q < kRecQuads; ++q) z[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // deepest entry any pixel of this wave / this tile needs
    uint32_t wave_last = last_contributor;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, m));
    if (lane == 0) atomicMax(&s_max, wave_last);
    __syncthreads();
    const uint32_t total = s_max;
    const uint32_t count = range.y - range.x;

    // entries behind the deepest contributor: zero records
    {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t e = total + tid; e < count; e += kBlock) {
            float4* o = inst_grads + (size_t)perm[range.x + e] * kRecQuads;
#pragma unroll
            for (int q = 0; q < kRecQuads; ++q) o[q] = zero;
        }
    }

    float T = T_final;
    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.f;
    float last_depth = 0.f, accum_depth_rec = 0.f, accum_alpha_rec = 0.f;
    float last_normal[3] = {0, 0, 0}, accum_normal_rec[3] = {0, 0, 0}, last_dL_dT = 0.f;

    const int rounds = (int)((total + kBlock - 1) / kBlock);
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kBlock;
        const uint32_t n = min((uint32_t)kBlock, total - rbase);
        if ((uint32_t)tid < n) {
            const float4* r = recs + (size_t)point_list[range.x + rbase + tid] * kRecQuads;
            float4 q[kRecQuads];
#pragma unroll
            for (int k = 0; k < kRecQuads; ++k) { q[k] = r[k]; s_q[k][tid] = q[k]; }
            s_mask[tid] = cull ? (uint8_t)quadrant_mask(q[0], q[1], q[2], Xc, Yc) : (uint8_t)0xF;
        }
        __syncthreads();
        if (wave_last > rbase) {
            // entries of this round that can touch this wave's quadrant and are not behind its deepest contributor
            const int cnt = build_wave_list(s_mask, s_list, wave, lane, n, wave_last - rbase);
            for (int idx = cnt - 1; idx >= 0; --idx) {
                const int j = (int)s_list[wave][idx];
                const uint32_t cidx = rbase + (uint32_t)j;  // 0-based contributor index
                Hit h;
                const float4 q0 = s_q[0][j], q1 = s_q[1][j], q2 = s_q[2][j];
                const bool valid = (cidx < last_contributor) && intersect(pxf, pyf, q0, q1, q2, h);
                if (__ballot(valid) == 0) continue;
                float v[20];
#pragma unroll
                for (int k = 0; k < 20; ++k) v[k] = 0.f;
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
                        v[16 + c] = w * gpix[c];
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
                        v[12 + c] = w * gN[c];
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
                        v[0] = -dkx; v[1] = -dky; v[2] = -dkz;
                        v[3] = -dlx; v[4] = -dly; v[5] = -dlz;
                        v[6] = pxf * dkx + pyf * dlx + dL_dz * h.sx;
                        v[7] = pxf * dky + pyf * dly + dL_dz * h.sy;
                        v[8] = pxf * dkz + pyf * dlz + dL_dz;
                    } else {
                        v[9] = dL_dG * (-h.G * kFilterInvSquare * h.dx);
                        v[10] = dL_dG * (-h.G * kFilterInvSquare * h.dy);
                        v[8] = dL_dz;
                    }
                    v[11] = h.G * dL_dalpha;
                }
                wave_reduce20(v);
                if ((lane & 15) == 0) {
                    float* a = &s_acc[j][5 * (lane >> 4)];
#pragma unroll
                    for (int k = 0; k < 5; ++k) atomicAdd(a + k, v[k]);  // ds_add_f32: 4 waves combine per splat
                }
            }
        }
        __syncthreads();
        // flush this round's records (one coalesced 80-B store per thread) and re-zero the accumulators
        if ((uint32_t)tid < n) {
            float4* acc = reinterpret_cast<float4*>(&s_acc[tid][0]);
            float4* o = inst_grads + (size_t)perm[range.x + rbase + tid] * kRecQuads;
#pragma unroll
            for (int q = 0; q < kRecQuads; ++q) { o[q] = acc[q]; acc[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        // (the next round's staging writes s_q rows; all reads of s_q finished at the barrier above)
    }
}

// launchers ---------------------------------------------------------------------------------------
hipError_t launch_render_forward(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                 float* out_color, float* out_allmap, float* final_T, uint32_t* n_contrib, int cull, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(render_forward_kernel, dim3(n_tiles), dim3(kBlock), 0, s, f, ranges, point_list, recs, out_color,
                       out_allmap, final_T, n_contrib, cull);
    return hipGetLastError();
}

hipError_t launch_render_backward(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                  const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                  const float* dL_dallmap, const uint32_t* perm, float4* inst_grads, int cull, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(render_backward_kernel, dim3(n_tiles), dim3(kBlock), 0, s, f, ranges, point_list, recs, final_T,
                       n_contrib, dL_dcolor, dL_dallmap, perm, inst_grads, cull);
    return hipGetLastError();
}

}  // namespace sr
