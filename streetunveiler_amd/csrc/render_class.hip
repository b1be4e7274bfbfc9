// render_class.hip -- the per-class distortion pass (second half): the reference's training iteration renders
// the same view once per semantic class with `render(..., semantic_filter_bit = 1 << k, reverse_semantic = True)` and uses only
// `rend_dist` of each [REF /root/reference/train.py:94-103] -- five full rasterizations (boolean-indexed inputs, K1..K8 each) for
// five distortion maps.  Here K1..K5 run ONCE on all Gaussians; the forward blend walks every tile list once and keeps one
// transmittance / distortion chain PER CLASS (a list entry belongs to exactly one class, which is wave-uniform, so the chain is
// picked by a uniform branch); the backward runs one wave per (tile, class) that skips the other classes' entries and stops at
// the class's deepest contributor; every (tile, Gaussian) duplicate gets at most one gradient record, so K8 is unchanged.
// Per class the arithmetic is exactly the class-filtered render's: same list order, same alpha / transmittance thresholds, same
// early termination -- a Gaussian of another class is to a class chain what it is to the reference's subset render: absent.
// The class id of a Gaussian travels in the first colour slot of its splat record (colors_precomp[:, 0]; there is no colour here).
#include "blend_common.h"

namespace sr {

constexpr int kClassMax = 8;

// QX x 1 quadrants per wave; SPLIT = 2: the reference's 16x16 tile as two 16x8 band waves (two pixels per lane), SPLIT = 1: the 8x8 /
// 16x8 / 32x8 tiles of BASELINE config 5's sweep, one wave per tile.
template <int NCLS, int QX, int SPLIT>
__global__ __launch_bounds__(kWave) void class_forward_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                              const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
                                                              float* __restrict__ out_dist,       // [NCLS, H, W]
                                                              float* __restrict__ cls_state,      // [NCLS, 3, H, W]: T_final, M1, M2
                                                              uint32_t* __restrict__ cls_last,    // [NCLS, H, W]: last contributor
                                                              uint32_t* __restrict__ tile_total,  // [tiles, NCLS]: deepest contributor of the class in the tile (zeroed by the caller)
                                                              uint16_t* __restrict__ hit_mask, int cull) {
    constexpr int QY = 1, NQ = QX;
    constexpr uint32_t kQuadMask = (1u << NQ) - 1u;
    __shared__ float4 s_e[entry_quads<3>()][kWave];
    const int lane = threadIdx.x;
    int tile = blockIdx.x, part = 0;
    if (SPLIT > 1) {   // the bands of a tile on one XCD (render.hip)
        const int xcd = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
        tile = (k / SPLIT) * kXcds + xcd; part = k % SPLIT;
        if (tile >= f.tiles_x * f.tiles_y) return;
    }
    tile = (int)tile_order[tile];
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8 * SPLIT) + part * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)((tile / f.tiles_x) * (QY * 8 * SPLIT) + QY * SPLIT * 4);
    const float yshift = (float)(part * (QY * 8) - QY * (SPLIT - 1) * 4);
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;
    const float yl = (float)(ly - QY * 4) + yshift;
    float xl[NQ];
    float T[NCLS][NQ], M1[NCLS][NQ], M2[NCLS][NQ], dist[NCLS][NQ];
    uint32_t lastc[NCLS][NQ];
    uint32_t done = 0, alive = 0;   // bit c * NQ + q: pixel (lane, q) finished for class c / some pixel of quadrant q still open for class c
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + q * 8 + lx, py = ty0 + ly;
        xl[q] = (float)(q * 8 + lx - QX * 4);
        const bool outside = !(px < f.W && py < f.H);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            T[c][q] = 1.f; M1[c][q] = M2[c][q] = dist[c][q] = 0.f; lastc[c][q] = 0;
            if (outside) done |= 1u << (c * NQ + q);
        }
        if (ballot64(!outside) != 0) {
#pragma unroll
            for (int c = 0; c < NCLS; ++c) alive |= 1u << (c * NQ + q);
        }
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nr[kRecQuads];
    if ((uint32_t)lane < n_total) load_record(recs, point_list[range.x + lane], nr);
    for (uint32_t base = 0; base < n_total && alive; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        uint32_t m = 0;
        bool has_class = false;
        if ((uint32_t)lane < n) {
            const float cls_f = nr[3].w;            // class id (colors_precomp[:, 0])
            const int ci = (int)cls_f;
            has_class = cls_f >= 0.f && ci < NCLS;
            m = stage_entry<QX, QY, 3>(nr, make_float4(0.f, 0.f, has_class ? (float)ci : -1.f, 0.f), zero4, Xc, Yc, cull & 1, s_e, lane, yshift);
            m = has_class ? (m & (alive >> (ci * NQ)) & kQuadMask) : 0u;
        }
        if (base + kWave + lane < n_total) load_record(recs, point_list[range.x + base + kWave + lane], nr);
        unsigned long long bits = ballot64(m != 0);
        unsigned long long hit[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) hit[q] = 0ull;
        while (bits) {
            const int j = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const int cj = __builtin_amdgcn_readfirstlane((int)e3.w);
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j) & (alive >> (cj * NQ)) & kQuadMask;
            if (!mj) continue;
            const uint32_t contributor = base + (uint32_t)j + 1u;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;  // wave-uniform
                Hit h;
                const bool ok = intersect(xl[q], yl, e0, e1, e2, e3, h);
#pragma unroll
                for (int c = 0; c < NCLS; ++c) {
                    if (cj != c) continue;        // wave-uniform: the entry's class picks the chain
                    const uint32_t bit = 1u << (c * NQ + q);
                    const bool valid = ok & !(done & bit);
                    if (ballot64(valid) == 0) continue;
                    hit[q] |= 1ull << j;
                    if (valid) {
                        const float test_T = T[c][q] * (1.f - h.alpha);
                        if (test_T < kTStop) {
                            done |= bit;  // this entry is NOT blended
                        } else {
                            const float w = h.alpha * T[c][q];
                            const float A = 1.f - T[c][q];
                            const float mm = kFN * (1.f - kNear * fast_rcp(h.depth));
                            dist[c][q] += (mm * mm * A + M2[c][q] - 2.f * mm * M1[c][q]) * w;
                            M1[c][q] += mm * w;
                            M2[c][q] += mm * mm * w;
                            T[c][q] = test_T;
                            lastc[c][q] = contributor;
                        }
                    }
                    if (ballot64(!(done & bit)) == 0) alive &= ~bit;
                }
            }
        }
        // (entry, quadrant) hit mask for the backward: one byte per band; only entries that carry a class are looked at there
        if ((uint32_t)lane < n && has_class) {
            uint32_t hm = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) hm |= (uint32_t)((hit[q] >> lane) & 1ull) << q;
            if (SPLIT == 2) reinterpret_cast<uint8_t*>(hit_mask)[2 * (size_t)(range.x + base + lane) + part] = (uint8_t)hm;
            else hit_mask[range.x + base + lane] = (uint16_t)hm;
        }
    }
    // entries behind the point where every chain of the band had closed keep whatever hit byte they had: the backward never gets
    // there (it stops at the class's deepest contributor)
    const size_t HW = (size_t)f.H * f.W;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        uint32_t deepest = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int px = tx0 + q * 8 + lx, py = ty0 + ly;
            if (px < f.W && py < f.H) {
                const size_t pix = (size_t)py * f.W + px;
                out_dist[c * HW + pix] = dist[c][q];
                cls_state[(c * 3 + 0) * HW + pix] = T[c][q]; cls_state[(c * 3 + 1) * HW + pix] = M1[c][q]; cls_state[(c * 3 + 2) * HW + pix] = M2[c][q];
                cls_last[c * HW + pix] = lastc[c][q];
            }
            deepest = max(deepest, lastc[c][q]);
        }
        deepest = wave_max_u32(deepest);
        if (lane == 0 && deepest) atomicMax(&tile_total[(size_t)tile * NCLS + c], deepest);
    }
}

// One wave per (tile, class): the blend backward of the class-filtered render with the distortion gradient as the only upstream
// gradient (no colour, depth, normal, alpha, median terms): psi = dLw, Z = sum_{k>i} w_k dLw_k  (render.hip, K7).
template <int NCLS, int QX, int QY>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 3)))
void class_backward_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list,
                           const float4* __restrict__ recs, const float* __restrict__ cls_state, const uint32_t* __restrict__ cls_last,
                           const uint32_t* __restrict__ tile_total, const float* __restrict__ dL_ddist, const uint16_t* __restrict__ hit_mask,
                           float4* __restrict__ inst_grads, uint8_t* __restrict__ written) {
    constexpr int NQ = QX * QY, kGQ = kGradQuads;
    __shared__ float4 s_e[entry_quads<3>()][kWave];
    __shared__ __attribute__((aligned(16))) float s_out[kWave][kGQ * 4];
    const int lane = threadIdx.x;
    const int tile = (int)tile_order[blockIdx.x / NCLS], cls = blockIdx.x % NCLS;
    const uint32_t total = tile_total[(size_t)tile * NCLS + cls];   // deepest list position any pixel of the tile needs for this class
    if (total == 0) return;
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)(ty0 + QY * 4);
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)f.H * f.W;
    const float xl0 = (float)(lx - QX * 4), yl0 = (float)(ly - QY * 4);
    float a0[NQ], a1[NQ], a2[NQ], T[NQ], Z[NQ];
    uint32_t lastc[NQ], quad_last[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        const bool inside = px < f.W && py < f.H;
        const size_t pix = inside ? (size_t)py * f.W + px : 0;
        const float T_final = inside ? cls_state[(cls * 3 + 0) * HW + pix] : 0.f;
        const float fin_D = inside ? cls_state[(cls * 3 + 1) * HW + pix] : 0.f, fin_D2 = inside ? cls_state[(cls * 3 + 2) * HW + pix] : 0.f;
        const float g_reg = inside ? dL_ddist[cls * HW + pix] : 0.f;
        lastc[q] = inside ? cls_last[cls * HW + pix] : 0u;
        a0[q] = (1.f - T_final) * g_reg; a1[q] = fin_D * g_reg; a2[q] = fin_D2 * g_reg;
        T[q] = T_final; Z[q] = 0.f;
        quad_last[q] = wave_max_u32(lastc[q]);
    }
    const int rounds = (int)((total + kWave - 1) / kWave);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nr[kRecQuads];
    uint32_t nhit = 0;
    auto fetch = [&](uint32_t pos) { const uint32_t gid = point_list[pos]; load_record(recs, gid, nr); nr[4].z = __uint_as_float(first_index(f, gid)); nhit = decode_hits<QX, QY>(hit_mask[pos]); };
    if ((uint32_t)((rounds - 1) * kWave + lane) < total) fetch(range.x + (rounds - 1) * kWave + lane);
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        uint32_t m = 0, slot = 0;
        if ((uint32_t)lane < n) {
            (void)stage_entry<QX, QY, 3>(nr, zero4, zero4, Xc, Yc, 0, s_e, lane);
            if (nr[3].w == (float)cls) {            // the other classes' entries are not there for this chain
                slot = emission_index(nr, __float_as_uint(nr[4].z), tile % f.tiles_x, tile / f.tiles_x, f);
                uint32_t need = 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) need |= (rbase + lane < quad_last[q]) ? (1u << q) : 0u;
                m = nhit & need;
            }
        }
        {
            float4* z = reinterpret_cast<float4*>(&s_out[lane][0]);
#pragma unroll
            for (int k = 0; k < kGQ; ++k) z[k] = zero4;
        }
        if (rd > 0) fetch(range.x + rbase - kWave + lane);
        unsigned long long bits = ballot64(m != 0);
        const unsigned long long wrote = bits;   // every entry with a forward hit gets a record (see K7)
        while (bits) {
            const int j = 63 - __clzll((long long)bits);
            bits &= ~(1ull << j);
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const uint32_t cidx = rbase + (uint32_t)j;
            float v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                v[k] = 0.f;
                if (k < 15) asm volatile("" : "+v"(v[k]));   // opaque zero for the live accumulators (see K7); 15..23 stay constant zero
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;
                Hit h;
                const float xq = xl0 + (float)((q % QX) * 8), yq = yl0 + (float)((q / QX) * 8);
                const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (cidx < lastc[q]);
                if (valid) {
                    const float Twx = e2.y, Twy = e2.z;
                    const float one_m_inv = fast_rcp(1.f - h.alpha);
                    T[q] *= one_m_inv;
                    const float w = h.alpha * T[q];
                    const float inv_depth = fast_rcp(h.depth);
                    const float m_d = kFN * (1.f - kNear * inv_depth);
                    const float dmd_dd = kFN * kNear * inv_depth * inv_depth;
#if SR_DETACH_WEIGHT
                    const float psi = 0.f;   // upstream DETACH_WEIGHT: no gradient through the blend weights (a distortion-only pass keeps d/dm)
#else
                    const float psi = a2[q] + m_d * (m_d * a0[q] - 2.f * a1[q]);
#endif
                    const float dL_dalpha = T[q] * psi - one_m_inv * Z[q];
                    Z[q] = fmaf(w, psi, Z[q]);
                    const float dL_dz = 2.f * w * (m_d * a0[q] - a1[q]) * dmd_dd;
                    const float dL_dG = e3.z * dL_dalpha;
                    v[14] += h.G * dL_dalpha;
                    if (h.use3d) {
                        const float gG = -dL_dG * h.G;
                        const float dpx = (gG * h.sx + dL_dz * Twx) * h.pz_inv, dpy = (gG * h.sy + dL_dz * Twy) * h.pz_inv;
                        const float dpz = -(dpx * h.sx + dpy * h.sy);
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
            {
                const float tot = wave_reduce24(v, lane);
                if (reduce24_holds_total(lane)) s_out[j][reduce24_index(lane)] = tot;
            }
        }
        if ((wrote >> lane) & 1ull) {
            const float4* accl = reinterpret_cast<const float4*>(&s_out[lane][0]);
            float4 acc[kGQ];
#pragma unroll
            for (int k = 0; k < kGQ; ++k) acc[k] = accl[k];
            const float4 centre = s_e[3][lane];   // (mx, my, ..) = the Gaussian's centre relative to the tile centre
            // (clamped into the image: the moments of a splat whose centre projects far off-screen are taken about the nearest image point)
            const float ox = -fminf(fmaxf(centre.x, -Xc), (float)(f.W - 1) - Xc), oy = -fminf(fmaxf(centre.y, -Yc), (float)(f.H - 1) - Yc);
            acc[0].w = fmaf(ox, acc[0].x, acc[0].w); acc[1].x = fmaf(ox, acc[0].y, acc[1].x); acc[1].y = fmaf(ox, acc[0].z, acc[1].y);
            acc[1].z = fmaf(oy, acc[0].x, acc[1].z); acc[1].w = fmaf(oy, acc[0].y, acc[1].w); acc[2].x = fmaf(oy, acc[0].z, acc[2].x);
            float4* o = inst_grads + (size_t)slot * kGQ;
#pragma unroll
            for (int k = 0; k < kGQ; ++k) o[k] = acc[k];
            written[slot] = 1;
        }
    }
}

hipError_t launch_class_forward(const FrameDev& f, int n_classes, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list,
                                const float4* recs, float* out_dist, float* cls_state, uint32_t* cls_last, uint32_t* tile_total, uint16_t* hit_mask,
                                int cull, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    const bool ref_tile = f.tile_w == 16 && f.tile_h == 16;
    if (!ref_tile && !(f.tile_h == 8 && (f.tile_w == 8 || f.tile_w == 16 || f.tile_w == 32))) return hipErrorInvalidValue;   // (32x16: eight pixels per lane)
    hipError_t e = hipMemsetAsync(tile_total, 0, sizeof(uint32_t) * (size_t)n_tiles * n_classes, s);
    if (e != hipSuccess) return e;
    const dim3 grid(ref_tile ? (n_tiles + kXcds - 1) / kXcds * kXcds * 2 : n_tiles);
#define SR_CF_SHAPE(N, QX, SPLIT) hipLaunchKernelGGL((class_forward_kernel<N, QX, SPLIT>), grid, dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, out_dist, cls_state, cls_last, tile_total, hit_mask, cull)
#define SR_CF(N) { if (ref_tile) SR_CF_SHAPE(N, 2, 2); else if (f.tile_w == 8) SR_CF_SHAPE(N, 1, 1); else if (f.tile_w == 16) SR_CF_SHAPE(N, 2, 1); else SR_CF_SHAPE(N, 4, 1); }
    switch (n_classes) { case 1: SR_CF(1); break; case 2: SR_CF(2); break; case 3: SR_CF(3); break; case 4: SR_CF(4); break;
                         case 5: SR_CF(5); break; case 6: SR_CF(6); break; default: return hipErrorInvalidValue; }
#undef SR_CF
#undef SR_CF_SHAPE
    return hipGetLastError();
}

hipError_t launch_class_backward(const FrameDev& f, int n_classes, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list,
                                 const float4* recs, const float* cls_state, const uint32_t* cls_last, const uint32_t* tile_total,
                                 const float* dL_ddist, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    const bool ref_tile = f.tile_w == 16 && f.tile_h == 16;
    if (!ref_tile && !(f.tile_h == 8 && (f.tile_w == 8 || f.tile_w == 16 || f.tile_w == 32))) return hipErrorInvalidValue;
#define SR_CB_SHAPE(N, QX, QY) hipLaunchKernelGGL((class_backward_kernel<N, QX, QY>), dim3(n_tiles * N), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, cls_state, cls_last, tile_total, dL_ddist, hit_mask, inst_grads, written)
#define SR_CB(N) { if (ref_tile) SR_CB_SHAPE(N, 2, 2); else if (f.tile_w == 8) SR_CB_SHAPE(N, 1, 1); else if (f.tile_w == 16) SR_CB_SHAPE(N, 2, 1); else SR_CB_SHAPE(N, 4, 1); }
    switch (n_classes) { case 1: SR_CB(1); break; case 2: SR_CB(2); break; case 3: SR_CB(3); break; case 4: SR_CB(4); break;
                         case 5: SR_CB(5); break; case 6: SR_CB(6); break; default: return hipErrorInvalidValue; }
#undef SR_CB
#undef SR_CB_SHAPE
    return hipGetLastError();
}

}  // namespace sr
