// render.hip -- K6 (per-tile front-to-back blend) and K7 (per-tile back-to-front backward), gfx950.
//
// Mapping: a wave64 owns QX x QY quadrants of 8x8 pixels, one pixel of each per lane -- lane l is pixel (l&7, l>>3) of every
// quadrant.  For the reference's 16x16 tile K7 is ONE wave per tile (2 x 2 quadrants, four pixels per lane: the gradient
// record of a (tile, Gaussian) pair wants one reduction over the whole tile) and K6 is TWO waves per tile (one per 16x8 band:
// fewer registers, more waves per SIMD -- these loops are latency-bound); other tile shapes: 1 / 2 / 4 / 8 quadrants per wave.
// A single-wave workgroup needs no barriers; the tile's depth-sorted splat list is staged through LDS 64 entries at a time
// (each lane gathers one 80-B packed record with five dwordx4 loads, the next round's records are already in flight while
// the current round is processed), and in the inner loop all 64 lanes read the same LDS address (broadcast ds_read_b128).
// No MFMA: there is no dense contraction in this path.
//
// Staging also rewrites each record into the form the inner loop wants (tile-local coordinates, which also
// removes the cancellation of the textbook k x l form):
//     p = k x l,  k = x Tw - Tu,  l = y Tw - Tv   ==   x (Tv x Tw) + y (Tw x Tu) + (Tu x Tv) = x A + y B + C
// and computes the quadrant mask (exact culling, see quadrant_mask).  The wave then walks only the
// entries whose mask is non-zero (scalar bit-scan over a ballot), and inside an entry only the quadrants
// whose bit is set.
//
// K7 keeps TWO floats of running state per pixel (T, Z) instead of the reference's 17: with the per-pixel upstream gradients
// folded in, the "colour/depth/normal accumulated behind" recurrences AND the distortion weight recurrence
// (last_dL_dT = dLw alpha + (1 - alpha) last_dL_dT, which equals sum_{k>i} w_k dLw_k / T_{i+1}) collapse into one suffix sum
//     Z_i = sum_{k>i} w_k psi_k - T_final (g_alpha - bg.g_rgb),   psi_k = rgb_k.g_rgb + depth_k g_depth + n_k.g_n + dLw_k,
//     dL/dalpha_i = T_i psi_i - Z_i / (1 - alpha_i)
// (algebraically identical to Appendix A.5), and the transMat gradient is accumulated as moments of dL/dp (S0, Sx, Sy, Z: see common.h)
// so the two cross products per (pixel, splat) pair of the textbook form run once per Gaussian in K8 instead.
// Per entry the 21 partial sums of all touched quadrants are added per lane, reduced across the wave ONCE
// (v_permlane32/16_swap + DPP: 60 VALU ops), and stored as one 96-B gradient record -- no atomics anywhere,
// deterministic.
//
// Behavioural contract: SURVEY.md Appendix A.4 / A.5; output channel order
// [REF /root/reference/gaussian_renderer/__init__.py:149-165].
#include "common.h"

namespace sr {

constexpr int kWave = 64;
constexpr int kXcds = 8;   // MI355X: 8 accelerator dies, workgroup i runs on XCD i % 8
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kFN = kFar / (kFar - kNear);

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// counters of the K6 counter variant (SrFrame.blend_counters, caller-owned, 8 x u64): [0] entries staged, [1] entries with a
// non-zero quadrant mask, [2] quadrant tests executed, [3] quadrant tests with >= 1 valid lane, [4] valid (pixel, entry) pairs,
// [5] / [6] tests with a valid pixel in rows 0-3 / rows 4-7 of the quadrant

// ---------------------------------------------------------------------------------------------
// Quadrant culling.  A list entry can only contribute to a pixel if alpha = min(0.99, opacity*G) >= 1/255,
// i.e. rho = min(rho3d, rho2d) <= thr = 2 ln(255 opacity).  {rho3d <= thr} is the image of the disc
// u^2+v^2 <= thr under the splat's homography -- an ellipse with dual conic C* = Q diag(thr,thr,-1) Q^T --
// and {rho2d <= thr} is a disc of radius sqrt(thr/2) around means2D.  The staging lane bounds that union by
// an octagon (support in directions x, y, x+y, x-y from the tangent-line equation l^T C* l = 0) and tests it
// against the tile's 8x8 quadrants (QX x QY of them; 2 x 2 for the reference's 16x16 tile).  (A second, nearly exact test in the splat's (u,v) plane removes another
// 9 % of the tests but costs more at staging than it saves -- measured, not kept.)  Entries dropped here are entries the per-pixel test would skip anyway (`continue` in Appendix A.4),
// so results are unchanged; the bound has 0.3 px / 1 % slack for float rounding and keeps the entry whenever
// anything is degenerate or NaN.  Inputs are tile-local (origin at
// the tile centre), which keeps the conic free of cancellation.
// ---------------------------------------------------------------------------------------------
template <int QX, int QY>
__device__ __forceinline__ uint32_t quadrant_mask(const float Tu[3], const float Tv[3], const float Tw[3], float mx, float my,
                                                  float opacity, float yshift) {
    constexpr uint32_t kAll = (1u << (QX * QY)) - 1u;
    float thr = 2.f * __logf(255.f * opacity);
    thr = thr * 1.01f + 0.01f;
    if (thr <= 0.f) return 0u;
    const float c22 = thr * (Tw[0] * Tw[0] + Tw[1] * Tw[1]) - Tw[2] * Tw[2];
    if (!(c22 < 0.f)) return kAll;  // the cutoff disc reaches the camera plane: unbounded footprint
    const float c00 = thr * (Tu[0] * Tu[0] + Tu[1] * Tu[1]) - Tu[2] * Tu[2];
    const float c01 = thr * (Tu[0] * Tv[0] + Tu[1] * Tv[1]) - Tu[2] * Tv[2];
    const float c11 = thr * (Tv[0] * Tv[0] + Tv[1] * Tv[1]) - Tv[2] * Tv[2];
    const float c02 = thr * (Tu[0] * Tw[0] + Tu[1] * Tw[1]) - Tu[2] * Tw[2];
    const float c12 = thr * (Tv[0] * Tw[0] + Tv[1] * Tw[1]) - Tv[2] * Tw[2];
    const float inv = fast_rcp(c22);
    const float r = __builtin_amdgcn_sqrtf(0.5f * thr);
    float lo[4], hi[4];
    const float QA[4] = {c00, c11, c00 + 2.f * c01 + c11, c00 - 2.f * c01 + c11};
    const float QB[4] = {c02, c12, c02 + c12, c02 - c12};
    const float ctr[4] = {mx, my, mx + my, mx - my};
    const float rad[4] = {r, r, r * 1.4142137f, r * 1.4142137f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float disc = QB[d] * QB[d] - QA[d] * c22;
        const float half = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * (-inv) * 1.01f;
        const float dc = QB[d] * inv;
        lo[d] = fminf(dc - half, ctr[d] - rad[d]);
        hi[d] = fmaxf(dc + half, ctr[d] + rad[d]);
    }
    // the wave's quadrants sit `yshift` below the local origin: shift the bounds instead of the (compile-time) rectangles
    lo[1] -= yshift; hi[1] -= yshift; lo[2] -= yshift; hi[2] -= yshift; lo[3] += yshift; hi[3] += yshift;
    const float m = 0.3f;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < QX * QY; ++q) {   // quadrant (q % QX, q / QX) of the tile, coordinates relative to the tile centre
        const float x0 = (float)((q % QX) * 8 - QX * 4) - m, x1 = (float)((q % QX) * 8 - QX * 4 + 7) + m;
        const float y0 = (float)((q / QX) * 8 - QY * 4) - m, y1 = (float)((q / QX) * 8 - QY * 4 + 7) + m;
        const bool out = lo[0] > x1 || hi[0] < x0 || lo[1] > y1 || hi[1] < y0 || lo[2] > x1 + y1 || hi[2] < x0 + y0 ||
                         lo[3] > x1 - y0 || hi[3] < x0 - y1;
        if (out) continue;
        mask |= 1u << q;
    }
    return mask;
}

// ---------------------------------------------------------------------------------------------
// Staged entry layout in LDS (struct-of-quads, s_e[quad][slot]):
//   e0 = A.xyz B.x | e1 = B.yz C.xy | e2 = C.z Tw.xyz | e3 = xy'.x xy'.y opacity c5
//   e4 = n.xyz c0  | e5 = c1 c2 c3 c4  | (9 channels only) e6 = c6 c7 c8 -
// with A = Tv' x Tw, B = Tw x Tu', C = Tu' x Tv' and ' = relative to the tile centre (Xc, Yc); c0..c2 = rgb.  SURVEY 8f N1:
// NC = 6 blends six precomputed channels (the two 3-channel one-hot passes of render_semantic as ONE pass), NC = 9 blends the
// SH colour AND six precomputed channels (render + render_semantic as one pass).  Channels 3.. come straight from the caller's
// [P,6] array at staging time: columns 3..5 for NC = 6 (columns 0..2 went through K1 into the record), all six for NC = 9.
// ---------------------------------------------------------------------------------------------
template <int NC> constexpr int entry_quads() { return NC == 9 ? 7 : 6; }

__device__ __forceinline__ float4 load_extra(const float* __restrict__ colors6, uint32_t gid, int first) {
    const float* c = colors6 + 6 * (size_t)gid + first;
    return make_float4(c[0], c[1], c[2], 0.f);
}

template <int QX, int QY, int NC>
__device__ __forceinline__ uint32_t stage_entry(const float4 (&q)[kRecQuads], const float4 ex, const float4 ey, float Xc, float Yc, int cull,
                                                float4 (*s_e)[kWave], int slot, float yshift = 0.f) {
    const float Tw[3] = {q[1].z, q[1].w, q[2].x};
    const float Tu[3] = {q[0].x - Xc * Tw[0], q[0].y - Xc * Tw[1], q[0].z - Xc * Tw[2]};
    const float Tv[3] = {q[0].w - Yc * Tw[0], q[1].x - Yc * Tw[1], q[1].y - Yc * Tw[2]};
    const float A[3] = {Tv[1] * Tw[2] - Tv[2] * Tw[1], Tv[2] * Tw[0] - Tv[0] * Tw[2], Tv[0] * Tw[1] - Tv[1] * Tw[0]};
    const float B[3] = {Tw[1] * Tu[2] - Tw[2] * Tu[1], Tw[2] * Tu[0] - Tw[0] * Tu[2], Tw[0] * Tu[1] - Tw[1] * Tu[0]};
    const float C[3] = {Tu[1] * Tv[2] - Tu[2] * Tv[1], Tu[2] * Tv[0] - Tu[0] * Tv[2], Tu[0] * Tv[1] - Tu[1] * Tv[0]};
    const float mx = q[2].y - Xc, my = q[2].z - Yc, opacity = q[2].w;
    s_e[0][slot] = make_float4(A[0], A[1], A[2], B[0]);
    s_e[1][slot] = make_float4(B[1], B[2], C[0], C[1]);
    s_e[2][slot] = make_float4(C[2], Tw[0], Tw[1], Tw[2]);
    s_e[3][slot] = make_float4(mx, my, opacity, ex.z);
    s_e[4][slot] = make_float4(q[3].x, q[3].y, q[3].z, q[4].x);
    s_e[5][slot] = make_float4(q[4].y, q[4].z, ex.x, ex.y);
    if (NC == 9) s_e[6][slot] = make_float4(ey.x, ey.y, ey.z, 0.f);
    return cull ? quadrant_mask<QX, QY>(Tu, Tv, Tw, mx, my, opacity, yshift) : (1u << (QX * QY)) - 1u;
}

struct Hit {
    float sx, sy, dx, dy, depth, G, alpha, pz_inv;
    bool use3d;
};

// Ray-splat intersection + alpha at tile-local pixel (xl, yl); branch-free, returns the validity predicate
// (the chain of `continue`s of Appendix A.4, with the same comparison senses so NaNs behave alike).
__device__ __forceinline__ bool intersect(float xl, float yl, const float4 e0, const float4 e1, const float4 e2, const float4 e3,
                                          Hit& h) {
    const float ppx = fmaf(xl, e0.x, fmaf(yl, e0.w, e1.z));
    const float ppy = fmaf(xl, e0.y, fmaf(yl, e1.x, e1.w));
    const float ppz = fmaf(xl, e0.z, fmaf(yl, e1.y, e2.x));
    h.pz_inv = fast_rcp(ppz);
    h.sx = ppx * h.pz_inv; h.sy = ppy * h.pz_inv;
    const float rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = e3.x - xl; h.dy = e3.y - yl;
    const float rho2d = kFilterInvSquare * (h.dx * h.dx + h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = h.use3d ? (h.sx * e2.y + h.sy * e2.z) + e2.w : e2.w;
    const float power = -0.5f * rho;
    h.G = __builtin_amdgcn_exp2f(power * kLog2e);
    h.alpha = fminf(kAlphaCap, e3.z * h.G);
    return (ppz != 0.f) & !(h.depth < kNear) & !(power > 0.f) & !(h.alpha < kAlphaFloor);
}

// Emission index of the duplicate (tile tx,ty ; Gaussian gid): duplicates are emitted per Gaussian, y-major /
// x-minor over its tile rectangle (same expressions as K1 / K3 -> same rectangle).
__device__ __forceinline__ uint32_t emission_index(const float4 (&q)[kRecQuads], uint32_t first, int tx, int ty, const FrameDev& f) {
    const float cx = q[2].y, cy = q[2].z, radius = q[4].w;
    int minx = (int)((cx - radius) * f.inv_tile_w), miny = (int)((cy - radius) * f.inv_tile_h);
    int maxx = (int)((cx + radius + (float)(f.tile_w - 1)) * f.inv_tile_w);
    minx = min(f.tiles_x, max(0, minx)); maxx = min(f.tiles_x, max(0, maxx));
    miny = min(f.tiles_y, max(0, miny));
    return first + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
}

__device__ __forceinline__ void load_record(const float4* __restrict__ recs, uint32_t gid, float4 (&q)[kRecQuads]) {
    const float4* r = recs + (size_t)gid * kRecQuads;
#pragma unroll
    for (int k = 0; k < kRecQuads; ++k) q[k] = r[k];
}

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
// SPLIT > 1: the parent tile (QX*8 wide, QY*8*SPLIT high -- the binning tile) is cut into SPLIT horizontal bands, one wave
// each, all walking the parent's list: fewer pixels per lane -> fewer registers -> more waves per SIMD, which is what these
// latency-bound loops want, at the price of staging every entry SPLIT times.
template <bool kStats, int NC, int QX, int QY, int SPLIT>
__global__ __launch_bounds__(kWave) void render_forward_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                                const uint32_t* __restrict__ point_list,
                                                                const float4* __restrict__ recs,
                                                                const float* __restrict__ extra,
                                                                float* __restrict__ out_color, float* __restrict__ out_allmap,
                                                                float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                                uint16_t* __restrict__ hit_mask, int cull, unsigned long long* __restrict__ g_stats) {
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    const int lane = threadIdx.x;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: the SPLIT bands of one tile take consecutive
    // slots of the SAME XCD so that the second band finds the tile's records in that L2 instead of fetching them again.
    int tile = blockIdx.x, part = 0;
    if (SPLIT > 1) {
        const int xcd = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
        tile = (k / SPLIT) * kXcds + xcd; part = k % SPLIT;
        if (tile >= f.tiles_x * f.tiles_y) return;
    }
    tile = (int)tile_order[tile];   // longest lists first (binning.hip tile_order_kernel)
    constexpr int NQ = QX * QY;   // 8x8 quadrants per wave = pixels per lane
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8 * SPLIT) + part * (QY * 8);
    // local origin = centre of the binning tile (shared with K7: identical staged values, identical decisions)
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)((tile / f.tiles_x) * (QY * 8 * SPLIT) + QY * SPLIT * 4);
    const float yshift = (float)(part * (QY * 8) - QY * (SPLIT - 1) * 4);   // this band's quadrants relative to that centre
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;

    float xl[NQ], yl[NQ];
    bool done[NQ];
    float T[NQ], C0[NQ], C1[NQ], C2[NQ], N0[NQ], N1[NQ], N2[NQ], Dsum[NQ], M1[NQ], M2[NQ], dist[NQ], med[NQ];
    float C3[NQ], C4[NQ], C5[NQ], C6[NQ], C7[NQ], C8[NQ];   // only live in the 6- / 9-channel variants
    uint32_t lastc[NQ], medc[NQ];
    uint32_t alive = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        xl[q] = (float)((q % QX) * 8 + lx - QX * 4); yl[q] = (float)((q / QX) * 8 + ly - QY * 4) + yshift;
        done[q] = !(px < f.W && py < f.H);
        T[q] = 1.f; C0[q] = C1[q] = C2[q] = N0[q] = N1[q] = N2[q] = 0.f;
        C3[q] = C4[q] = C5[q] = C6[q] = C7[q] = C8[q] = 0.f;
        Dsum[q] = M1[q] = M2[q] = dist[q] = med[q] = 0.f;
        lastc[q] = 0; medc[q] = 0xFFFFFFFFu;
        if (__ballot(!done[q]) != 0) alive |= 1u << q;
    }

    float4 nr[kRecQuads], nx = make_float4(0.f, 0.f, 0.f, 0.f), ny = nx;
    if ((uint32_t)lane < n_total) {
        const uint32_t gid = point_list[range.x + lane];
        load_record(recs, gid, nr);
        if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
    }
    for (uint32_t base = 0; base < n_total && alive; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        uint32_t m = 0;
        if ((uint32_t)lane < n) m = stage_entry<QX, QY, NC>(nr, nx, ny, Xc, Yc, cull & 1, s_e, lane, yshift);
        if (base + kWave + lane < n_total) {
            const uint32_t gid = point_list[range.x + base + kWave + lane];
            load_record(recs, gid, nr);
            if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
        }
        unsigned long long bits = __ballot((m & alive) != 0);
        if (kStats && lane == 0) { atomicAdd(&g_stats[0], (unsigned long long)n); atomicAdd(&g_stats[1], (unsigned long long)__popcll(bits)); }
        unsigned long long hit[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) hit[q] = 0ull;  // scalar: bit j of hit[q] = entry j reached a pixel of quadrant q
        while (bits) {
            const int j = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j) & alive;
            if (!mj) continue;
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const uint32_t contributor = base + (uint32_t)j + 1u;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;  // wave-uniform
                Hit h;
                const bool valid = intersect(xl[q], yl[q], e0, e1, e2, e3, h) & !done[q];
                if (kStats) {
                    const unsigned long long vb = __ballot(valid);
                    if (lane == 0) { atomicAdd(&g_stats[2], 1ull); if (vb) atomicAdd(&g_stats[3], 1ull); atomicAdd(&g_stats[4], (unsigned long long)__popcll(vb));
                                     if (vb & 0xFFFFFFFFull) atomicAdd(&g_stats[5], 1ull); if (vb >> 32) atomicAdd(&g_stats[6], 1ull); }
                }
                if (__ballot(valid) == 0) continue;
                hit[q] |= 1ull << j;
                const float4 e4 = s_e[4][j], e5 = s_e[5][j];
                if (valid) {
                    const float test_T = T[q] * (1.f - h.alpha);
                    if (test_T < kTStop) {
                        done[q] = true;  // this entry is NOT blended
                    } else {
                        const float w = h.alpha * T[q];
                        const float A = 1.f - T[q];
                        const float mm = kFN * (1.f - kNear * fast_rcp(h.depth));
                        dist[q] += (mm * mm * A + M2[q] - 2.f * mm * M1[q]) * w;
                        Dsum[q] += h.depth * w;
                        M1[q] += mm * w;
                        M2[q] += mm * mm * w;
                        if (T[q] > 0.5f) { med[q] = h.depth; medc[q] = contributor; }
                        N0[q] += e4.x * w; N1[q] += e4.y * w; N2[q] += e4.z * w;
                        C0[q] += e4.w * w; C1[q] += e5.x * w; C2[q] += e5.y * w;
                        if (NC >= 6) { C3[q] += e5.z * w; C4[q] += e5.w * w; C5[q] += e3.w * w; }
                        if (NC == 9) { const float4 e6 = s_e[6][j]; C6[q] += e6.x * w; C7[q] += e6.y * w; C8[q] += e6.z * w; }
                        T[q] = test_T;
                        lastc[q] = contributor;
                    }
                }
                if (__ballot(!done[q]) == 0) alive &= ~(1u << q);
            }
        }
        // exact (entry, quadrant) hit mask for the backward: K7 visits only the pairs that reached a pixel here
        if ((uint32_t)lane < n) {
            uint32_t hm = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) hm |= (uint32_t)((hit[q] >> lane) & 1ull) << q;
            // 16 bits per list entry; two-band tiles: low byte = quadrants of the upper band, high byte = lower band
            if (SPLIT == 2) reinterpret_cast<uint8_t*>(hit_mask)[2 * (size_t)(range.x + base + lane) + part] = (uint8_t)hm;
            else if (QY == 2) hit_mask[range.x + base + lane] = (uint16_t)((hm & ((1u << QX) - 1u)) | ((hm >> QX) << 8));
            else hit_mask[range.x + base + lane] = (uint16_t)hm;
        }
    }
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            final_T[pix] = T[q]; final_T[HW + pix] = M1[q]; final_T[2 * HW + pix] = M2[q];
            n_contrib[pix] = lastc[q]; n_contrib[HW + pix] = medc[q];
            out_color[pix] = C0[q] + T[q] * bg0;
            out_color[HW + pix] = C1[q] + T[q] * bg1;
            out_color[2 * HW + pix] = C2[q] + T[q] * bg2;
            if (NC >= 6) {
                out_color[3 * HW + pix] = C3[q] + T[q] * f.bg[3];
                out_color[4 * HW + pix] = C4[q] + T[q] * f.bg[4];
                out_color[5 * HW + pix] = C5[q] + T[q] * f.bg[5];
            }
            if (NC == 9) {
                out_color[6 * HW + pix] = C6[q] + T[q] * f.bg[6];
                out_color[7 * HW + pix] = C7[q] + T[q] * f.bg[7];
                out_color[8 * HW + pix] = C8[q] + T[q] * f.bg[8];
            }
            out_allmap[pix] = Dsum[q];
            out_allmap[HW + pix] = 1.f - T[q];
            out_allmap[2 * HW + pix] = N0[q]; out_allmap[3 * HW + pix] = N1[q]; out_allmap[4 * HW + pix] = N2[q];
            out_allmap[5 * HW + pix] = med[q];
            out_allmap[6 * HW + pix] = dist[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wave-level transpose-reduction of 24 values per lane (gfx950):
//   fold across the two 32-lane halves with v_permlane32_swap (value k <-> k+12), across row pairs with
//   v_permlane16_swap (k <-> k+6), then a 4-step DPP row rotation sum.  60 VALU ops for 24 values (a plain
//   butterfly needs 144 cross-lane ops).  Afterwards every lane of 16-lane row g holds, in v[0..5], the
//   64-lane totals of values 6g .. 6g+5.
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
// value of lane ^ 4, without the LDS crossbar: a row rotation by 12 (= left by 4) into the lanes whose bit 2 is clear (DPP
// banks 0 and 2) and by 4 into the others (banks 1 and 3); row_ror:n delivers lane (l - n) mod 16
__device__ __forceinline__ float dpp_xor4(float x) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x12C, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x124, 0xf, 0xA, false);
    return __int_as_float(t);
}
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0x128>(x);  // row_ror:8
    x += dpp_mov<0x124>(x);  // row_ror:4
    x += dpp_mov<0x122>(x);  // row_ror:2
    x += dpp_mov<0x121>(x);  // row_ror:1
    return x;
}
// Returns the 64-lane total of ONE value per lane: lane l ends up with value index
//   6*(l>>4) + 3*bit3(l) + (bit2(l) ? 2 : bit1(l)), valid unless bit2 and bit1 are both set; bit0 is a replica.
// Folds all the way down (24 -> 12 -> 6 -> 3 -> 2 -> 1 values per lane), so the expensive cross-lane steps
// shrink geometrically: 18 swap-adds + 7 in-row exchanges instead of 18 swap-adds + 24 DPP adds
// (tools/ubench/reduce_ubench.hip: 416 vs 633 SIMD cycles per reduction).
__device__ __forceinline__ float wave_reduce24(float (&v)[24], int lane) {
#pragma unroll
    for (int k = 0; k < 12; ++k) fold32(v[k], v[k + 12]);
#pragma unroll
    for (int k = 0; k < 6; ++k) fold16(v[k], v[k + 6]);
    const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float keep = h8 ? v[k + 3] : v[k], send = h8 ? v[k] : v[k + 3];
        v[k] = keep + dpp_mov<0x128>(send);  // row_ror:8
    }
    {
        const float keep0 = h4 ? v[2] : v[0], send0 = h4 ? v[0] : v[2];
        const float keep1 = h4 ? 0.f : v[1], send1 = h4 ? v[1] : 0.f;
        v[0] = keep0 + dpp_xor4(send0);
        v[1] = keep1 + dpp_xor4(send1);
    }
    {
        const float keep = h2 ? v[1] : v[0], send = h2 ? v[0] : v[1];
        v[0] = keep + dpp_mov<0x4E>(send);   // quad_perm:[2,3,0,1] = lane ^ 2
    }
    return v[0] + dpp_mov<0xB1>(v[0]);       // quad_perm:[1,0,3,2] = lane ^ 1
}

// 64-lane totals of three more values (the 9-channel variant): afterwards every lane of 16-lane row r holds the total of value r
// (row 3: zero).  Two half folds, one row-pair fold, one in-row sum.
__device__ __forceinline__ float wave_reduce3(float a, float b, float c) {
    float d = 0.f;
    fold32(a, b); fold32(c, d);   // a: lanes < 32 hold a's half sums, lanes >= 32 b's;  c: c's | zeros
    fold16(a, c);                 // rows 0..3: a, c, b, zero
    return row_sum16(a);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, m));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}

template <int QX, int QY>
__device__ __forceinline__ uint32_t decode_hits(uint16_t h) {   // see the hit_mask store in K6
    // two-band tiles (16x16, 32x16): low byte = the QX quadrant bits of the upper band, high byte = those of the lower band
    return QY == 2 ? (((uint32_t)h & ((1u << QX) - 1u)) | ((((uint32_t)h >> 8) & ((1u << QX) - 1u)) << QX)) : (uint32_t)h;
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
// Output: one 96-B gradient record per (tile, Gaussian) duplicate, stored at the duplicate's EMISSION index
// (first[gid] + its index inside the Gaussian's tile rectangle; first[gid] rides in slot 15 of the splat record), where the
// records of one Gaussian are contiguous; K8 sums them.  Only entries with a contributing pixel get a record, and a 1 in
// `written[]` at the same index (zeroed per call).  Gradient record slots: see common.h.
// Register budget: three waves per SIMD (<= 168 VGPRs) wherever the per-pixel state allows it -- the loop is latency-bound
// at two -- i.e. up to four pixels per lane with three colour channels.
template <int NC, int QX, int QY>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(QX * QY <= 4 && NC == 3 ? 3 : 1, QX * QY <= 4 && NC == 3 ? 3 : 8)))
void render_backward_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                                 const uint32_t* __restrict__ point_list,
                                                                 const float4* __restrict__ recs,
                                                                 const float* __restrict__ extra,
                                                                 const float* __restrict__ final_T,
                                                                 const uint32_t* __restrict__ n_contrib,
                                                                 const float* __restrict__ dL_dcolor,
                                                                 const float* __restrict__ dL_dallmap,
                                                                 const uint16_t* __restrict__ hit_mask,
                                                                 float4* __restrict__ inst_grads, uint8_t* __restrict__ written) {
    constexpr int kGQ = NC == 9 ? kGradQuads + 1 : kGradQuads;   // quads per gradient record (27 values with 9 channels)
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    __shared__ __attribute__((aligned(16))) float s_out[kWave][kGQ * 4];
    const int lane = threadIdx.x;
    const int tile = (int)tile_order[blockIdx.x];   // longest lists first
    constexpr int NQ = QX * QY;   // 8x8 quadrants per tile = pixels per lane
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)(ty0 + QY * 4);
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const uint32_t count = range.y - range.x;
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];

    // per-pixel constants (upstream gradients folded with the forward's final accumulators) and state
    const float xl0 = (float)(lx - QX * 4), yl0 = (float)(ly - QY * 4);   // tile-local pixel of quadrant 0; quadrant q adds 8*(q%QX, q/QX)
    float gr[NQ], gg[NQ], gb[NQ], gn0[NQ], gn1[NQ], gn2[NQ], g_depth[NQ], g_median[NQ], a0[NQ], a1[NQ], a2[NQ];
    float gc3[NQ], gc4[NQ], gc5[NQ], gc6[NQ], gc7[NQ], gc8[NQ];   // only live in the 6- / 9-channel variants
    uint32_t lastc[NQ], medc[NQ], quad_last[NQ];
    float T[NQ], Z[NQ];
    uint32_t total = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        const bool inside = px < f.W && py < f.H;
        const size_t pix = inside ? (size_t)py * f.W + px : 0;
        const float T_final = inside ? final_T[pix] : 0.f;
        const float fin_D = inside ? final_T[HW + pix] : 0.f, fin_D2 = inside ? final_T[2 * HW + pix] : 0.f;
        lastc[q] = inside ? n_contrib[pix] : 0u;
        medc[q] = inside ? n_contrib[HW + pix] : 0u;
        gr[q] = inside ? dL_dcolor[pix] : 0.f; gg[q] = inside ? dL_dcolor[HW + pix] : 0.f; gb[q] = inside ? dL_dcolor[2 * HW + pix] : 0.f;
        g_depth[q] = inside ? dL_dallmap[pix] : 0.f;
        const float g_accum = inside ? dL_dallmap[HW + pix] : 0.f;
        gn0[q] = inside ? dL_dallmap[2 * HW + pix] : 0.f; gn1[q] = inside ? dL_dallmap[3 * HW + pix] : 0.f; gn2[q] = inside ? dL_dallmap[4 * HW + pix] : 0.f;
        g_median[q] = inside ? dL_dallmap[5 * HW + pix] : 0.f;
        const float g_reg = inside ? dL_dallmap[6 * HW + pix] : 0.f;
        float bg_dot = bg0 * gr[q] + bg1 * gg[q] + bg2 * gb[q];
        gc3[q] = gc4[q] = gc5[q] = gc6[q] = gc7[q] = gc8[q] = 0.f;
        if (NC >= 6) {
            gc3[q] = inside ? dL_dcolor[3 * HW + pix] : 0.f; gc4[q] = inside ? dL_dcolor[4 * HW + pix] : 0.f; gc5[q] = inside ? dL_dcolor[5 * HW + pix] : 0.f;
            bg_dot += f.bg[3] * gc3[q] + f.bg[4] * gc4[q] + f.bg[5] * gc5[q];
        }
        if (NC == 9) {
            gc6[q] = inside ? dL_dcolor[6 * HW + pix] : 0.f; gc7[q] = inside ? dL_dcolor[7 * HW + pix] : 0.f; gc8[q] = inside ? dL_dcolor[8 * HW + pix] : 0.f;
            bg_dot += f.bg[6] * gc6[q] + f.bg[7] * gc7[q] + f.bg[8] * gc8[q];
        }
        a0[q] = (1.f - T_final) * g_reg; a1[q] = fin_D * g_reg; a2[q] = fin_D2 * g_reg;
        T[q] = T_final; Z[q] = -T_final * (g_accum - bg_dot);   // the background / alpha term rides in the suffix sum
        quad_last[q] = wave_max_u32(lastc[q]);  // deepest entry any pixel of quadrant q needs (uniform)
        total = max(total, quad_last[q]);
    }

    // entries behind the deepest contributor of the tile are never looked at: they get no record and keep a clear `written` flag
    const int rounds = (int)((total + kWave - 1) / kWave);
    float4 nr[kRecQuads], nx = make_float4(0.f, 0.f, 0.f, 0.f), ny = nx;
    uint32_t nhit = 0;
    if (rounds > 0 && (uint32_t)((rounds - 1) * kWave + lane) < total) {
        const uint32_t pos = range.x + (rounds - 1) * kWave + lane;
        const uint32_t gid = point_list[pos];
        load_record(recs, gid, nr);
        if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
        nhit = decode_hits<QX, QY>(hit_mask[pos]);
    }
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        uint32_t m = 0, slot = 0;
        if ((uint32_t)lane < n) {
            (void)stage_entry<QX, QY, NC>(nr, nx, ny, Xc, Yc, 0, s_e, lane);
            m = nhit;   // (entry, quadrant) pairs that reached a pixel in the forward: exact, no culling test needed here
            slot = emission_index(nr, __float_as_uint(nr[3].w), tile % f.tiles_x, tile / f.tiles_x, f);
            uint32_t need = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) need |= (rbase + lane < quad_last[q]) ? (1u << q) : 0u;
            m &= need;
        }
        {
            float4* z = reinterpret_cast<float4*>(&s_out[lane][0]);
#pragma unroll
            for (int k = 0; k < kGQ; ++k) z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (rd > 0) {  // next round is always full
            const uint32_t pos = range.x + rbase - kWave + lane;
            const uint32_t gid = point_list[pos];
            load_record(recs, gid, nr);
            if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
            nhit = decode_hits<QX, QY>(hit_mask[pos]);
        }
        unsigned long long bits = __ballot(m != 0);
        unsigned long long wrote = 0ull;   // scalar: entries of this round that got a contribution (only those get a record)
        while (bits) {
            const int j = 63 - __clzll((long long)bits);
            bits &= ~(1ull << j);
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const uint32_t cidx = rbase + (uint32_t)j;  // 0-based contributor index
            float v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                v[k] = 0.f;
                asm volatile("" : "+v"(v[k]));   // opaque zero: every quadrant block accumulates in place (no phi copies of constants)
            }
            float w6 = 0.f, w7 = 0.f, w8 = 0.f;   // colour channels 6..8 (9-channel variant)
            bool any = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;  // wave-uniform
                Hit h;
                const float xq = xl0 + (float)((q % QX) * 8), yq = yl0 + (float)((q / QX) * 8);
                const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (cidx < lastc[q]);
                if (__ballot(valid) == 0) continue;
                any = true;
                const float4 e4 = s_e[4][j], e5 = s_e[5][j];
                if (valid) {
                    const float Twx = e2.y, Twy = e2.z;
                    const float one_m_inv = fast_rcp(1.f - h.alpha);
                    T[q] *= one_m_inv;                 // transmittance in front of this entry
                    const float w = h.alpha * T[q];
                    float phi = fmaf(e4.w, gr[q], fmaf(e5.x, gg[q], fmaf(e5.y, gb[q], fmaf(h.depth, g_depth[q],
                                fmaf(e4.x, gn0[q], fmaf(e4.y, gn1[q], e4.z * gn2[q]))))));
                    if (NC >= 6) phi = fmaf(e5.z, gc3[q], fmaf(e5.w, gc4[q], fmaf(e3.w, gc5[q], phi)));
                    if (NC == 9) { const float4 e6 = s_e[6][j]; phi = fmaf(e6.x, gc6[q], fmaf(e6.y, gc7[q], fmaf(e6.z, gc8[q], phi))); }
                    const float inv_depth = fast_rcp(h.depth);
                    const float m_d = kFN * (1.f - kNear * inv_depth);
                    const float dmd_dd = kFN * kNear * inv_depth * inv_depth;
                    const float psi = phi + (a2[q] + m_d * (m_d * a0[q] - 2.f * a1[q]));
                    const float dL_dalpha = T[q] * psi - one_m_inv * Z[q];
                    Z[q] = fmaf(w, psi, Z[q]);
                    float dL_dz = 2.f * w * (m_d * a0[q] - a1[q]) * dmd_dd + w * g_depth[q];
                    if (cidx == medc[q] - 1u) dL_dz += g_median[q];
                    const float dL_dG = e3.z * dL_dalpha;
                    v[18] += w * gr[q]; v[19] += w * gg[q]; v[20] += w * gb[q];
                    if (NC >= 6) { v[21] += w * gc3[q]; v[22] += w * gc4[q]; v[23] += w * gc5[q]; }
                    if (NC == 9) { w6 += w * gc6[q]; w7 += w * gc7[q]; w8 += w * gc8[q]; }
                    v[15] += w * gn0[q]; v[16] += w * gn1[q]; v[17] += w * gn2[q];
                    v[14] += h.G * dL_dalpha;
                    if (h.use3d) {
                        const float gG = -dL_dG * h.G;
                        const float dpx = (gG * h.sx + dL_dz * Twx) * h.pz_inv, dpy = (gG * h.sy + dL_dz * Twy) * h.pz_inv;
                        const float dpz = -(dpx * h.sx + dpy * h.sy);
                        // moments of dL/dp in tile-local pixel coordinates (shifted to global ones when the record is written);
                        // the cross products happen once per Gaussian in K8
                        v[0] += dpx; v[1] += dpy; v[2] += dpz;
                        v[3] = fmaf(xq, dpx, v[3]); v[4] = fmaf(xq, dpy, v[4]); v[5] = fmaf(xq, dpz, v[5]);
                        v[6] = fmaf(yq, dpx, v[6]); v[7] = fmaf(yq, dpy, v[7]); v[8] = fmaf(yq, dpz, v[8]);
                        v[9] = fmaf(dL_dz, h.sx, v[9]); v[10] = fmaf(dL_dz, h.sy, v[10]); v[11] += dL_dz;
                    } else {
                        const float gG = -dL_dG * h.G * kFilterInvSquare;
                        v[12] = fmaf(gG, h.dx, v[12]);
                        v[13] = fmaf(gG, h.dy, v[13]);
                        v[11] += dL_dz;
                    }
                }
            }
            if (any) {
                wrote |= 1ull << j;
                const float tot = wave_reduce24(v, lane);
                if ((lane & 1) == 0 && (lane & 6) != 6)
                    s_out[j][6 * (lane >> 4) + ((lane & 8) ? 3 : 0) + ((lane & 4) ? 2 : ((lane >> 1) & 1))] = tot;
                if (NC == 9) {
                    const float t3 = wave_reduce3(w6, w7, w8);   // row 0: channel 6, row 1: channel 8, row 2: channel 7
                    if ((lane & 15) == 0 && lane < 48) s_out[j][24 + (lane == 0 ? 0 : (lane == 16 ? 2 : 1))] = t3;
                }
            }
        }
        // flush this round's records: one 96-B store per lane whose entry got a contribution
        if ((wrote >> lane) & 1ull) {
            const float4* accl = reinterpret_cast<const float4*>(&s_out[lane][0]);
            float4 acc[kGQ];
#pragma unroll
            for (int k = 0; k < kGQ; ++k) acc[k] = accl[k];
            // Sx, Sy from tile-local to global pixel coordinates: sum (Xc + xl) dp = Xc S0 + sum xl dp
            acc[0].w = fmaf(Xc, acc[0].x, acc[0].w); acc[1].x = fmaf(Xc, acc[0].y, acc[1].x); acc[1].y = fmaf(Xc, acc[0].z, acc[1].y);
            acc[1].z = fmaf(Yc, acc[0].x, acc[1].z); acc[1].w = fmaf(Yc, acc[0].y, acc[1].w); acc[2].x = fmaf(Yc, acc[0].z, acc[2].x);
            float4* o = inst_grads + (size_t)slot * kGQ;
#pragma unroll
            for (int k = 0; k < kGQ; ++k) o[k] = acc[k];
            written[slot] = 1;   // K8 reads this 1-B flag (zeroed per call) before it touches the record
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decision dump (test infrastructure of the parity bars, not part of the operator): for every list entry of every tile and
// every pixel of the tile, whether the ray-splat test of K6 / K7 accepts the pair (`valid`: the chain of skips of Appendix A.4
// up to alpha >= 1/255, WITHOUT the pixel's saturation state) and which path it takes (`use3d`: rho3d <= rho2d).  Same staging,
// same `intersect`, same local origin as the blend kernels => the same bits they act on.  One wave per tile, lane l = pixel
// (l & 7, l >> 3) of each 8x8 quadrant; out[(list position) * QX*QY + quadrant] = 64-bit ballot.
// ---------------------------------------------------------------------------------------------
template <int QX, int QY>
__global__ __launch_bounds__(kWave) void pair_decisions_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                               const float4* __restrict__ recs, unsigned long long* __restrict__ valid_bits,
                                                               unsigned long long* __restrict__ use3d_bits) {
    __shared__ float4 s_e[entry_quads<3>()][kWave];
    constexpr int NQ = QX * QY;
    const int lane = threadIdx.x, tile = blockIdx.x;
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)(ty0 + QY * 4);
    const float xl0 = (float)((lane & 7) - QX * 4), yl0 = (float)((lane >> 3) - QY * 4);
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t base = 0; base < n_total; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        if ((uint32_t)lane < n) {
            float4 nr[kRecQuads];
            load_record(recs, point_list[range.x + base + lane], nr);
            (void)stage_entry<QX, QY, 3>(nr, zero4, zero4, Xc, Yc, 0, s_e, lane);
        }
        for (uint32_t j = 0; j < n; ++j) {
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                Hit h;
                const bool valid = intersect(xl0 + (float)((q % QX) * 8), yl0 + (float)((q / QX) * 8), e0, e1, e2, e3, h);
                const unsigned long long vb = __ballot(valid), ub = __ballot(h.use3d);
                if (lane == 0) { valid_bits[(size_t)(range.x + base + j) * NQ + q] = vb; use3d_bits[(size_t)(range.x + base + j) * NQ + q] = ub; }
            }
        }
    }
}

// launchers ---------------------------------------------------------------------------------------
// Tile shapes (BASELINE config 5's sweep): the reference's 16x16 plus 8x8, 16x8, 32x8, 32x16 = QX x QY quadrants of 8x8
// pixels, i.e. 1 / 2 / 4 / 8 pixels per lane.  Only the reference shape carries the 6- / 9-channel and counter variants.
#define SR_FOR_TILE_SHAPE(F)                                                    \
    if (f.tile_w == 16 && f.tile_h == 16) { F(2, 2); }                          \
    else if (f.tile_w == 8 && f.tile_h == 8) { F(1, 1); }                       \
    else if (f.tile_w == 16 && f.tile_h == 8) { F(2, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 8) { F(4, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 16) { F(4, 2); }                     \
    else return hipErrorInvalidValue;

// flags: bit 0 = quadrant culling on (SR_FLAG_NO_QUADRANT_CULL clear), bit 1 = counter variant (counters != NULL)
hipError_t launch_render_forward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                 const float* extra, float* out_color, float* out_allmap, float* final_T, uint32_t* n_contrib,
                                 uint16_t* hit_mask, int flags, unsigned long long* counters, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    const dim3 block(kWave);
    const int cull = flags & 1;
    const bool count = (flags & 2) != 0 && counters != nullptr;
#define SR_LAUNCH_FWD(STATS, NCH, QX, QY, SPLIT)                                                                                  \
    hipLaunchKernelGGL((render_forward_kernel<STATS, NCH, QX, QY, SPLIT>),                                                          \
                       dim3(SPLIT > 1 ? (n_tiles + kXcds - 1) / kXcds * kXcds * SPLIT : n_tiles), block, 0, s, f, \
                       ranges, tile_order, point_list, recs, extra, out_color, out_allmap, final_T, n_contrib, hit_mask, cull, counters)
    if (f.tile_w == 16 && f.tile_h == 16) {
        // the reference's tile: two 16x8 band waves per tile (the counter variant stays whole so that it counts each entry once)
        if (f.colors == 9) { SR_LAUNCH_FWD(false, 9, 2, 1, 2); }
        else if (f.colors == 6) { if (count) SR_LAUNCH_FWD(true, 6, 2, 2, 1); else SR_LAUNCH_FWD(false, 6, 2, 1, 2); }
        else if (count)         SR_LAUNCH_FWD(true, 3, 2, 2, 1);
        else                    SR_LAUNCH_FWD(false, 3, 2, 1, 2);
    } else {
        if (f.colors != 3) return hipErrorInvalidValue;
        if (f.tile_w == 32 && f.tile_h == 16) { SR_LAUNCH_FWD(false, 3, 4, 1, 2); }   // two 32x8 band waves per tile
        else {
#define SR_FWD_SHAPE(QX, QY) SR_LAUNCH_FWD(false, 3, QX, QY, 1)
            SR_FOR_TILE_SHAPE(SR_FWD_SHAPE)
#undef SR_FWD_SHAPE
        }
    }
#undef SR_LAUNCH_FWD
    return hipGetLastError();
}

hipError_t launch_render_backward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                  const float* extra, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                  const float* dL_dallmap, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
#define SR_LAUNCH_BWD(NCH, QX, QY)                                                                                                          \
    hipLaunchKernelGGL((render_backward_kernel<NCH, QX, QY>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, extra, final_T, \
                       n_contrib, dL_dcolor, dL_dallmap, hit_mask, inst_grads, written)
    if (f.tile_w == 16 && f.tile_h == 16) {
        if (f.colors == 9) SR_LAUNCH_BWD(9, 2, 2); else if (f.colors == 6) SR_LAUNCH_BWD(6, 2, 2); else SR_LAUNCH_BWD(3, 2, 2);
    } else {
        if (f.colors != 3) return hipErrorInvalidValue;
#define SR_BWD_SHAPE(QX, QY) SR_LAUNCH_BWD(3, QX, QY)
        SR_FOR_TILE_SHAPE(SR_BWD_SHAPE)
#undef SR_BWD_SHAPE
    }
#undef SR_LAUNCH_BWD
    return hipGetLastError();
}

hipError_t launch_pair_decisions(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                 unsigned long long* valid_bits, unsigned long long* use3d_bits, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
#define SR_DEC_SHAPE(QX, QY) hipLaunchKernelGGL((pair_decisions_kernel<QX, QY>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, point_list, recs, valid_bits, use3d_bits)
    SR_FOR_TILE_SHAPE(SR_DEC_SHAPE)
#undef SR_DEC_SHAPE
    return hipGetLastError();
}

}  // namespace sr
