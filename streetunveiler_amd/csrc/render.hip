// render.hip -- K6 (per-tile front-to-back blend); K7 (per-tile back-to-front backward) lives in render_bwd.hip; gfx950.
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
#include "blend_common.h"

#ifndef SR_K6_COLOURS_UP_FRONT
#define SR_K6_COLOURS_UP_FRONT 1
#endif
namespace sr {

// exact (entry, quadrant) hit mask for the backward: K7 visits only the pairs that reached a pixel in the forward.  16 bits per list entry;
// two-band tiles: low byte = quadrants of the upper band, high byte = lower band (decode_hits, blend_common.h)
template <int QX, int QY, int SPLIT>
__device__ __forceinline__ void store_hit_mask(uint16_t* __restrict__ hit_mask, uint32_t pos, int part, uint32_t hm) {
    if (SPLIT == 2) reinterpret_cast<uint8_t*>(hit_mask)[2 * (size_t)pos + part] = (uint8_t)hm;
    else if (QY == 2) hit_mask[pos] = (uint16_t)((hm & ((1u << QX) - 1u)) | ((hm >> QX) << 8));
    else hit_mask[pos] = (uint16_t)hm;
}

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
// SPLIT > 1: the parent tile (QX*8 wide, QY*8*SPLIT high -- the binning tile) is cut into SPLIT horizontal bands, one wave
// each, all walking the parent's list: fewer pixels per lane -> fewer registers -> more waves per SIMD, which is what these
// latency-bound loops want, at the price of staging every entry SPLIT times.
template <bool kStats, int NC, int QX, int QY, int SPLIT>
__device__ __forceinline__ void render_forward_body(float4 (*s_e)[kWave], const FrameDev& f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                    const uint32_t* __restrict__ point_list, const float4* __restrict__ recs, const float* __restrict__ extra,
                                                    float* __restrict__ out_color, float* __restrict__ out_allmap, float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask, int cull, unsigned long long* __restrict__ g_stats) {
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
    const int yshift_px = part * (QY * 8) - QY * (SPLIT - 1) * 4;   // this band's quadrants relative to that centre
    const float yshift = (float)yshift_px;
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;

    float xl[NQ], yl[NQ];
    // T[q] > 0: the pixel is live.  T[q] < 0: done (saturated, or outside the image) -- |T[q]| is its final transmittance.  (A separate
    // flag per pixel costs a byte compare, two moves and a four-instruction all-done test per quadrant test.)
    float T[NQ], C0[NQ], C1[NQ], C2[NQ], N0[NQ], N1[NQ], N2[NQ], Dsum[NQ], M1[NQ], M2[NQ], dist[NQ], med[NQ];
    float C3[NQ], C4[NQ], C5[NQ], C6[NQ], C7[NQ], C8[NQ];   // only live in the 6- / 9-channel variants
    uint32_t lastc[NQ], medc[NQ];
    uint32_t alive = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        xl[q] = (float)((q % QX) * 8 + lx - QX * 4); yl[q] = (float)((q / QX) * 8 + ly - QY * 4) + yshift;
        T[q] = (px < f.W && py < f.H) ? 1.f : -1.f; C0[q] = C1[q] = C2[q] = N0[q] = N1[q] = N2[q] = 0.f;
        C3[q] = C4[q] = C5[q] = C6[q] = C7[q] = C8[q] = 0.f;
        Dsum[q] = M1[q] = M2[q] = dist[q] = med[q] = 0.f;
        lastc[q] = 0; medc[q] = 0xFFFFFFFFu;
        if (ballot64(T[q] > 0.f) != 0) alive |= 1u << q;
    }

    // Memory pipeline of the walk (round 6): the list entry (gid) is requested TWO rounds ahead, the record it addresses one round ahead, and
    // the hit masks of a round are stored behind the staging of the next one -- so the only wait of a round (for the records, at its top)
    // is for operations issued a whole round earlier.  Before, `load_record(recs, point_list[..])` made every round wait for the list
    // entry's round trip, and the wait for the records also waited for the hit-mask store issued right in front of it.
    float4 nr[kRecQuads], nx = make_float4(0.f, 0.f, 0.f, 0.f), ny = nx;
    uint32_t gid_ahead = 0;
    if ((uint32_t)lane < n_total) {
        const uint32_t gid = point_list[range.x + lane];
        load_record18(recs, gid, nr);   // (depth and radius, the record's last two floats, are not staged by the forward)
        if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
    }
    if (kWave + (uint32_t)lane < n_total) gid_ahead = point_list[range.x + kWave + lane];
    // the hit masks of the round that ended last, not stored yet: they wait as the round's scalar ballots (no vector register across the staging)
    unsigned long long hit_prev[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) hit_prev[q] = 0ull;
    uint32_t n_prev = 0, base_prev = 0;
    for (uint32_t base = 0; base < n_total && alive; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        uint32_t m = 0;
        int ys = yshift_px;
        asm volatile("" : "+s"(ys));   // converted again in every round: one v_cvt per 64 entries instead of a register held across the walk
        uint32_t cells16 = 0;
        wait_vector_memory();   // every operation in flight was issued a round ago; on every path, so that nothing later in the round waits again
        if ((uint32_t)lane < n) m = stage_entry<QX, QY, NC>(nr, nx, ny, Xc, Yc, cull & 1, s_e, lane, (float)ys, kStats && QX == 2 && QY == 2 ? &cells16 : nullptr);
        // (the record loads are the LAST memory operations of the round: nothing needs a temporary register while their 18 destinations
        // are in flight -- at the 80-register budget the allocator otherwise moves half-arrived quads around, behind an s_waitcnt)
        const uint32_t gid = gid_ahead;
        if (hit_mask && (uint32_t)lane < n_prev) {
            uint32_t hm = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) hm |= (uint32_t)((hit_prev[q] >> lane) & 1ull) << q;
            store_hit_mask<QX, QY, SPLIT>(hit_mask, range.x + base_prev + lane, part, hm);
        }
        if (base + 2 * kWave + lane < n_total) gid_ahead = point_list[range.x + base + 2 * kWave + lane];
        __builtin_amdgcn_sched_barrier(0);
        if (base + kWave + lane < n_total) {
            load_record18(recs, gid, nr);   // (depth and radius, the record's last two floats, are not staged by the forward)
            if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t alive_at_round_start = alive;
        unsigned long long bits = ballot64((m & alive) != 0);
        if (kStats && lane == 0) { atomicAdd(&g_stats[0], (unsigned long long)n); atomicAdd(&g_stats[1], (unsigned long long)__popcll(bits)); }
        unsigned long long hit[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) hit[q] = 0ull;  // scalar: bit j of hit[q] = entry j reached a pixel of quadrant q
        uint32_t cell_round[NQ][4];   // (counter variant) entries of this round with a valid pixel in 4x4 cell c of quadrant q
#pragma unroll
        for (int q = 0; q < NQ; ++q) { cell_round[q][0] = cell_round[q][1] = cell_round[q][2] = cell_round[q][3] = 0u; }
        while (bits) {
            const int j = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j) & alive;
            if (!mj) continue;
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            // 6 / 9 channels (four waves per SIMD, registers to spare): normal and colours of the entry fetched here, once per entry, beside its geometry --
            // inside the quadrant test every test with a hit waited for its own LDS round trip (render_backward_kernel does the same).  The three-channel
            // kernel lives on its sixth wave at 80 registers and keeps the loads where they are.
            constexpr bool kColoursUpFront = SR_K6_COLOURS_UP_FRONT && NC != 3 && !kStats;
            float4 u4 = make_float4(0.f, 0.f, 0.f, 0.f), u5 = u4, u6 = u4;
            if (kColoursUpFront) { u4 = s_e[4][j]; u5 = s_e[5][j]; if (NC == 9) u6 = s_e[6][j]; }
            const uint32_t contributor = base + (uint32_t)j + 1u;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;  // wave-uniform
                Hit h;
                const bool valid = intersect(xl[q], yl[q], e0, e1, e2, e3, h) & (T[q] > 0.f);
                if (kStats) {
                    const unsigned long long vb = ballot64(valid);
                    if (lane == 0) { atomicAdd(&g_stats[2], 1ull); if (vb) atomicAdd(&g_stats[3], 1ull); atomicAdd(&g_stats[4], (unsigned long long)__popcll(vb));
                                     if (vb & 0xFFFFFFFFull) atomicAdd(&g_stats[5], 1ull); if (vb >> 32) atomicAdd(&g_stats[6], 1ull); }
                    // bit of pixel (x, y) of the quadrant = 8 y + x: the four 4x4 cells
                    cell_round[q][0] += (vb & 0x000000000F0F0F0Full) != 0; cell_round[q][1] += (vb & 0x00000000F0F0F0F0ull) != 0;
                    cell_round[q][2] += (vb & 0x0F0F0F0F00000000ull) != 0; cell_round[q][3] += (vb & 0xF0F0F0F000000000ull) != 0;
                    if (NQ == 4) {   // is the oriented-box cell mask conservative?  an exact hit in a cell whose box bit is clear is counted in [14]
                        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cells16, j) >> (16 + 4 * q);
                        const uint32_t exact = ((vb & 0x000000000F0F0F0Full) != 0 ? 1u : 0u) | ((vb & 0x00000000F0F0F0F0ull) != 0 ? 2u : 0u) |
                                               ((vb & 0x0F0F0F0F00000000ull) != 0 ? 4u : 0u) | ((vb & 0xF0F0F0F000000000ull) != 0 ? 8u : 0u);
                        if (lane == 0 && (exact & ~cj & 15u)) atomicAdd(&g_stats[14], (unsigned long long)__popc(exact & ~cj & 15u));
                    }
                }
                if (ballot64(valid) == 0) continue;
                hit[q] |= 1ull << j;
                const float4 e4 = kColoursUpFront ? u4 : s_e[4][j], e5 = kColoursUpFront ? u5 : s_e[5][j];
                if (valid) {
                    const float test_T = T[q] * (1.f - h.alpha);
                    const bool go = !(test_T < kTStop);   // else: done, and this entry is NOT blended
                    if (go) {
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
                        if (NC == 9) { const float4 e6 = kColoursUpFront ? u6 : s_e[6][j]; C6[q] += e6.x * w; C7[q] += e6.y * w; C8[q] += e6.z * w; }
                        lastc[q] = contributor;
                    }
                    T[q] = go ? test_T : -T[q];
                }
                if (ballot64(T[q] > 0.f) == 0) alive &= ~(1u << q);
            }
        }
        if (kStats && lane == 0) {   // [7]: entries with a contributing pixel somewhere in the tile = gradient records K7 will write
            unsigned long long any = 0ull;
#pragma unroll
            for (int q = 0; q < NQ; ++q) any |= hit[q];
            atomicAdd(&g_stats[7], (unsigned long long)__popcll(any));
            // what a mapping of 16-lane rows on independent entries (one 4x4 cell per row, four rows = one quadrant per wave, exact
            // cell culling) would execute for this round of 64 entries: the busiest row of every quadrant sets its wave's step count
            unsigned long long cells = 0ull, steps = 0ull;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                cells += cell_round[q][0] + cell_round[q][1] + cell_round[q][2] + cell_round[q][3];
                steps += max(max(cell_round[q][0], cell_round[q][1]), max(cell_round[q][2], cell_round[q][3]));
            }
            atomicAdd(&g_stats[8], cells); atomicAdd(&g_stats[9], steps);
        }
        if (kStats && NQ == 4) {   // ... and with the (entry, cell) pairs a conservative octagon-vs-cell test at staging would hand to the rows
            unsigned long long kept = 0ull, steps = 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t busiest = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t k = (uint32_t)__popcll(ballot64(((cells16 >> (4 * q + c)) & 1u) != 0u && ((m >> q) & 1u) != 0u && ((alive_at_round_start >> q) & 1u) != 0u));
                    kept += k; busiest = max(busiest, k);
                }
                steps += busiest;
            }
            if (lane == 0) { atomicAdd(&g_stats[10], kept); atomicAdd(&g_stats[11], steps); }
            kept = 0ull; steps = 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t busiest = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t k = (uint32_t)__popcll(ballot64(((cells16 >> (16 + 4 * q + c)) & 1u) != 0u && ((m >> q) & 1u) != 0u && ((alive_at_round_start >> q) & 1u) != 0u));
                    kept += k; busiest = max(busiest, k);
                }
                steps += busiest;
            }
            if (lane == 0) { atomicAdd(&g_stats[12], kept); atomicAdd(&g_stats[13], steps); }
        }
        // exact (entry, quadrant) hit mask for the backward: K7 visits only the pairs that reached a pixel here
        if (hit_mask) {   // (NULL with SR_FLAG_FORWARD_ONLY: no backward will read it)
#pragma unroll
            for (int q = 0; q < NQ; ++q) hit_prev[q] = hit[q];
            n_prev = n; base_prev = base;   // stored behind the next round's staging, or behind the walk
        }
    }
    if (hit_mask && (uint32_t)lane < n_prev) {
        uint32_t hm = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) hm |= (uint32_t)((hit_prev[q] >> lane) & 1ull) << q;
        store_hit_mask<QX, QY, SPLIT>(hit_mask, range.x + base_prev + lane, part, hm);
    }
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];
    int lane_again = threadIdx.x;
    asm volatile("" : "+v"(lane_again));   // the pixel coordinates are recomputed here instead of living in registers across the list walk
    const int lx2 = lane_again & 7, ly2 = lane_again >> 3;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx2, py = ty0 + (q / QX) * 8 + ly2;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            const float Tq = fabsf(T[q]);
            if (final_T) {   // the backward's per-pixel state (NULL with SR_FLAG_FORWARD_ONLY)
                final_T[pix] = Tq; final_T[HW + pix] = M1[q]; final_T[2 * HW + pix] = M2[q];
                n_contrib[pix] = lastc[q]; n_contrib[HW + pix] = medc[q];
            }
            out_color[pix] = C0[q] + Tq * bg0;
            out_color[HW + pix] = C1[q] + Tq * bg1;
            out_color[2 * HW + pix] = C2[q] + Tq * bg2;
            if (NC >= 6) {
                out_color[3 * HW + pix] = C3[q] + Tq * f.bg[3];
                out_color[4 * HW + pix] = C4[q] + Tq * f.bg[4];
                out_color[5 * HW + pix] = C5[q] + Tq * f.bg[5];
            }
            if (NC == 9) {
                out_color[6 * HW + pix] = C6[q] + Tq * f.bg[6];
                out_color[7 * HW + pix] = C7[q] + Tq * f.bg[7];
                out_color[8 * HW + pix] = C8[q] + Tq * f.bg[8];
            }
            out_allmap[pix] = Dsum[q];
            out_allmap[HW + pix] = 1.f - Tq;
            out_allmap[2 * HW + pix] = N0[q]; out_allmap[3 * HW + pix] = N1[q]; out_allmap[4 * HW + pix] = N2[q];
            out_allmap[5 * HW + pix] = med[q];
            out_allmap[6 * HW + pix] = dist[q];
        }
    }
}

template <bool kStats, int NC, int QX, int QY, int SPLIT>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(!kStats && NC == 3 && QX == 2 && QY == 1 && SPLIT == 2 ? 6 : 1, !kStats && NC == 3 && QX == 2 && QY == 1 && SPLIT == 2 ? 6 : 8)))
void render_forward_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list,
                           const float4* __restrict__ recs, const float* __restrict__ extra, float* __restrict__ out_color, float* __restrict__ out_allmap,
                           float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask, int cull,
                           unsigned long long* __restrict__ g_stats) {
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    render_forward_body<kStats, NC, QX, QY, SPLIT>(s_e, f, ranges, tile_order, point_list, recs, extra, out_color, out_allmap, final_T, n_contrib, hit_mask, cull, g_stats);
}

// ---------------------------------------------------------------------------------------------
// K6, row-mapped (three colour channels): the wave's four 16-lane rows are the four 4x4 cells of a quadrant, and every row walks ITS
// OWN list -- the entries of the round whose octagon reaches its cell -- so a wave step serves four different entries, one per row,
// instead of one entry on 64 lanes of which a thin splat uses a dozen.  Same staging, same `intersect`, same per-pixel sequence of
// operations as render_forward_kernel: bit-identical images, state and hit masks.  The rows' entry indices travel packed in one SGPR
// (next set bit of the row's 64-bit mask: scalar unit), every lane extracts its row's byte and reads the staged entry at ITS address.
// ---------------------------------------------------------------------------------------------
// kCells (round 6, the row-mapped BACKWARD's input: render_backward_rows_kernel): the hit masks are kept per (entry, 4x4 CELL) instead of per
// (entry, quadrant) -- a row IS a cell here, so the exact bit costs nothing: bit 4 q + c of the band's byte = cell c of quadrant q of the band
// (16x16 tile: low byte = upper band, high byte = lower band, i.e. bit 4 q + c of the 16-bit word with q = the tile's quadrant).  The
// bytes wait in LDS as s_hit[q][entry][cell] (one word per (q, entry)).
template <int QX, int QY, int SPLIT, bool kCells = false>
__device__ __forceinline__ void render_forward_rows_body(float4 (*s_e)[kWave], uint8_t (*s_hit)[kCells ? 4 * kWave : kWave], const FrameDev& f, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list,
                                                         const float4* __restrict__ recs, float* __restrict__ out_color, float* __restrict__ out_allmap,
                                                         float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask) {
    constexpr int NC = 3;
    const int lane = threadIdx.x;
    int tile = blockIdx.x, part = 0;
    if (SPLIT > 1) {
        const int xcd = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
        tile = (k / SPLIT) * kXcds + xcd; part = k % SPLIT;
        if (tile >= f.tiles_x * f.tiles_y) return;
    }
    tile = (int)tile_order[tile];
    constexpr int NQ = QX * QY;
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8 * SPLIT) + part * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)((tile / f.tiles_x) * (QY * 8 * SPLIT) + QY * SPLIT * 4);
    const int yshift_px = part * (QY * 8) - QY * (SPLIT - 1) * 4;
    const float yshift = (float)yshift_px;
    // row r = lane / 16 <-> cell (r & 1, r >> 1) of the quadrant; lane % 16 <-> pixel (l & 3, l >> 2) of the cell
    const int lx = ((lane >> 4) & 1) * 4 + (lane & 3), ly = (lane >> 5) * 4 + ((lane >> 2) & 3);
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;

    const float xl0 = (float)(lx - QX * 4), yl0 = (float)(ly - QY * 4) + yshift;   // quadrant 0; quadrant q adds 8 (q % QX, q / QX)
    float T[NQ], C0[NQ], C1[NQ], C2[NQ], N0[NQ], N1[NQ], N2[NQ], Dsum[NQ], M1[NQ], M2[NQ], dist[NQ], med[NQ];
    uint32_t lastc[NQ], medc[NQ];
    uint32_t alive = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        T[q] = (px < f.W && py < f.H) ? 1.f : -1.f; C0[q] = C1[q] = C2[q] = N0[q] = N1[q] = N2[q] = 0.f;
        Dsum[q] = M1[q] = M2[q] = dist[q] = med[q] = 0.f;
        lastc[q] = 0; medc[q] = 0xFFFFFFFFu;
        if (ballot64(T[q] > 0.f) != 0) alive |= 1u << q;
    }
    float4 nr[kRecQuads];
    const float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the 18 floats this kernel stages -- everything of the record but depth and radius, which sit in its last two slots -- are prefetched a
    // round ahead: 18 registers across the walk keep the kernel at 80 VGPRs = six waves per SIMD.  Round 3 prefetched four quads and fetched
    // the fifth at staging time, a round later: by then its line had left the L1 and often the L2 -- 13.7 M extra L1->L2 requests and
    // +0.33 GB of raw FETCH_SIZE per launch, tools/notes_round4_measured.md)
    // (memory pipeline as in render_forward_body: list entries two rounds ahead, records one, hit masks stored behind the next staging)
    uint32_t gid_ahead = 0;
    if ((uint32_t)lane < n_total) load_record18(recs, point_list[range.x + lane], nr);
    if (kWave + (uint32_t)lane < n_total) gid_ahead = point_list[range.x + kWave + lane];
    uint32_t hm_prev = 0, n_prev = 0, base_prev = 0;
    for (uint32_t base = 0; base < n_total && alive; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        int ys = yshift_px;
        asm volatile("" : "+s"(ys));
        uint32_t cm = 0;   // this lane's ENTRY: bit 4 q + c = its octagon reaches cell c of quadrant q
        wait_vector_memory();
        if ((uint32_t)lane < n) (void)stage_entry<QX, QY, NC>(nr, nx, nx, Xc, Yc, 1, s_e, lane, (float)ys, nullptr, &cm);
        const uint32_t gid = gid_ahead;
        if (hit_mask && (uint32_t)lane < n_prev) store_hit_mask<QX, QY, SPLIT>(hit_mask, range.x + base_prev + lane, part, hm_prev);
        if (base + 2 * kWave + lane < n_total) gid_ahead = point_list[range.x + base + 2 * kWave + lane];
        __builtin_amdgcn_sched_barrier(0);
        if (base + kWave + lane < n_total) load_record18(recs, gid, nr);   // (last: see render_forward_body)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { if (kCells) reinterpret_cast<uint32_t*>(&s_hit[q][0])[lane] = 0u; else s_hit[q][lane] = 0; }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!(alive & (1u << q))) continue;
            const float xq = xl0 + (float)((q % QX) * 8), yq = yl0 + (float)((q / QX) * 8);
            // this lane's ROW's list for the quadrant: the scalar unit serves four SIMDs and would be the bottleneck of a walk on scalar
            // masks (measured: 60 scalar instructions per step), so every lane keeps its row's mask in a register pair
            unsigned long long brow;
            {
                const unsigned long long b0 = ballot64(((cm >> (4 * q)) & 1u) != 0u), b1 = ballot64(((cm >> (4 * q + 1)) & 1u) != 0u),
                                         b2 = ballot64(((cm >> (4 * q + 2)) & 1u) != 0u), b3 = ballot64(((cm >> (4 * q + 3)) & 1u) != 0u);
                const int row = lane >> 4;
                brow = row == 0 ? b0 : (row == 1 ? b1 : (row == 2 ? b2 : b3));
            }
            while (ballot64(brow != 0ull) != 0ull) {
                const bool act = brow != 0ull;
                const uint32_t j = act ? (uint32_t)__builtin_ctzll(brow) : 0u;
                brow &= brow - 1ull;
                const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
                Hit h;
                const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (T[q] > 0.f) & act;
                if (ballot64(valid) == 0ull) continue;
                const float4 e4 = s_e[4][j], e5 = s_e[5][j];
                if (valid) {
                    if (kCells) s_hit[q][4 * j + (lane >> 4)] = 1; else s_hit[q][j] = 1;   // (every valid lane of the row stores the same byte)
                    const uint32_t contributor = base + j + 1u;
                    const float test_T = T[q] * (1.f - h.alpha);
                    const bool go = !(test_T < kTStop);
                    if (go) {
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
                        lastc[q] = contributor;
                    }
                    T[q] = go ? test_T : -T[q];
                }
                // (a row whose sixteen pixels are done could drop the rest of its list: the test costs more per step than the steps it saves)
                if (ballot64(T[q] > 0.f) == 0ull) { alive &= ~(1u << q); break; }
            }
        }
        if (hit_mask) {
            uint32_t hm = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (kCells) {   // four 0 / 1 bytes -> a nibble: the products land on distinct bits (21 + c), no carries
                    const uint32_t wd = reinterpret_cast<const uint32_t*>(&s_hit[q][0])[lane];
                    hm |= (((wd * 0x00204081u) >> 21) & 15u) << (4 * q);
                } else hm |= (uint32_t)s_hit[q][lane] << q;
            }
            hm_prev = hm; n_prev = n; base_prev = base;
        }
    }
    if (hit_mask && (uint32_t)lane < n_prev) store_hit_mask<QX, QY, SPLIT>(hit_mask, range.x + base_prev + lane, part, hm_prev);
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];
    int lane_again = threadIdx.x;
    asm volatile("" : "+v"(lane_again));   // the pixel coordinates are recomputed here instead of living in registers across the list walk
    const int lx2 = ((lane_again >> 4) & 1) * 4 + (lane_again & 3), ly2 = (lane_again >> 5) * 4 + ((lane_again >> 2) & 3);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx2, py = ty0 + (q / QX) * 8 + ly2;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            const float Tq = fabsf(T[q]);
            if (final_T) {   // the backward's per-pixel state (NULL with SR_FLAG_FORWARD_ONLY)
                final_T[pix] = Tq; final_T[HW + pix] = M1[q]; final_T[2 * HW + pix] = M2[q];
                n_contrib[pix] = lastc[q]; n_contrib[HW + pix] = medc[q];
            }
            out_color[pix] = C0[q] + Tq * bg0;
            out_color[HW + pix] = C1[q] + Tq * bg1;
            out_color[2 * HW + pix] = C2[q] + Tq * bg2;
            out_allmap[pix] = Dsum[q];
            out_allmap[HW + pix] = 1.f - Tq;
            out_allmap[2 * HW + pix] = N0[q]; out_allmap[3 * HW + pix] = N1[q]; out_allmap[4 * HW + pix] = N2[q];
            out_allmap[5 * HW + pix] = med[q];
            out_allmap[6 * HW + pix] = dist[q];
        }
    }
}

template <int QX, int QY, int SPLIT, bool kCells = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(QX * QY <= 2 ? 6 : 1, QX * QY <= 2 ? 6 : 8)))
void render_forward_rows_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list,
                                const float4* __restrict__ recs, float* __restrict__ out_color, float* __restrict__ out_allmap, float* __restrict__ final_T,
                                uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask) {
    static_assert(!kCells || (QX == 2 && QY == 1 && SPLIT == 2), "cell-granular hit masks: the 16x16 tile's band waves (eight cells per band byte)");
    __shared__ float4 s_e[entry_quads<3>()][kWave];
    __shared__ __attribute__((aligned(4))) uint8_t s_hit[QX * QY][kCells ? 4 * kWave : kWave];   // (entry, quadrant [, cell]) reached a pixel: the backward's exact visit list
    render_forward_rows_body<QX, QY, SPLIT, kCells>(s_e, s_hit, f, ranges, tile_order, point_list, recs, out_color, out_allmap, final_T, n_contrib, hit_mask);
}


// ---------------------------------------------------------------------------------------------
// K6, cooperative form (round 6): FOUR waves per 16x16 tile, one per 8x8 quadrant (one pixel per lane), one workgroup, ONE staging of every
// entry (each wave stages 16 of the round's 64 into the shared s_e).  For frames with few tiles -- the reference's own `-r 4` runs:
// 480x320 = 600 tiles [REF /root/reference/README.md:195-207] -- where two band waves per tile leave most wave slots empty.  Same lists, same
// staging arithmetic, same `intersect`, the same per-pixel sequence of operations as render_forward_kernel: images, per-pixel state and hit
// masks are bit-identical to the band kernel's (a test requires it).  Three colour channels; culling always on.
// MEASURED, and therefore only run when asked for (SR_FLAG_COOP_BACKWARD in sr_forward_render): 1.5 M Gaussians at 480x320 0.481 ms against
// the band kernel's 0.487 (8x8 tiles: 0.37); C3 0.968 against 0.852 -- the three barriers per round make every quadrant wait for the slowest
// one, which costs what the single staging saves.  The BACKWARD's cooperative form is the one that pays on small frames (render_bwd.hip).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(4 * kWave) void render_forward_coop_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                                          const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
                                                                          float* __restrict__ out_color, float* __restrict__ out_allmap, float* __restrict__ final_T,
                                                                          uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask) {
    constexpr int NC = 3, kStage = kWave / 4;
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    __shared__ uint32_t s_m[kWave];
    __shared__ uint8_t s_hitq[4][kWave];
    __shared__ uint32_t s_alive[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = (int)tile_order[blockIdx.x];
    const int tx0 = (tile % f.tiles_x) * 16, ty0 = (tile / f.tiles_x) * 16;
    const float Xc = (float)(tx0 + 8), Yc = (float)(ty0 + 8);
    const int lx = lane & 7, ly = lane >> 3, qx = w & 1, qy = w >> 1;
    const int px = tx0 + qx * 8 + lx, py = ty0 + qy * 8 + ly;
    const float xl = (float)(qx * 8 + lx - 8), yl = (float)(qy * 8 + ly - 8);
    const uint2 range = ranges[tile];
    const uint32_t n_total = range.y - range.x;
    float T = (px < f.W && py < f.H) ? 1.f : -1.f;   // > 0: live; < 0: done, |T| = final transmittance (render_forward_body)
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, Dsum = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    uint32_t lastc = 0, medc = 0xFFFFFFFFu;
    if (lane == 0) s_alive[w] = 0u;
    __syncthreads();
    if (ballot64(T > 0.f) != 0ull && lane == 0) s_alive[w] = 1u;
    __syncthreads();
    const bool stager = lane < kStage;
    const int se = w * kStage + lane;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nr[kRecQuads];
    uint32_t gid_ahead = 0;
    if (stager && (uint32_t)se < n_total) load_record18(recs, point_list[range.x + se], nr);
    if (stager && kWave + (uint32_t)se < n_total) gid_ahead = point_list[range.x + kWave + se];
    uint32_t hm_prev = 0, n_prev = 0, base_prev = 0;
    for (uint32_t base = 0; base < n_total; base += kWave) {
        const uint32_t alive = (s_alive[0] ? 1u : 0u) | (s_alive[1] ? 2u : 0u) | (s_alive[2] ? 4u : 0u) | (s_alive[3] ? 8u : 0u);   // (uniform: read behind a barrier)
        if (!alive) break;
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        wait_vector_memory();
        if (stager) {
            uint32_t m = 0;
            if ((uint32_t)se < n) m = stage_entry<2, 2, NC>(nr, zero4, zero4, Xc, Yc, 1, s_e, se) & alive;
            s_m[se] = m;
            const uint32_t gid = gid_ahead;
            if (hit_mask && (uint32_t)se < n_prev) hit_mask[range.x + base_prev + se] = (uint16_t)((hm_prev & 3u) | ((hm_prev >> 2) << 8));   // (decode_hits<2, 2>: a byte per quadrant row)
            if (base + 2 * kWave + se < n_total) gid_ahead = point_list[range.x + base + 2 * kWave + se];
            __builtin_amdgcn_sched_barrier(0);
            if (base + kWave + se < n_total) load_record18(recs, gid, nr);
            __builtin_amdgcn_sched_barrier(0);
        }
        s_hitq[w][lane] = 0;
        __syncthreads();
        unsigned long long bits = ballot64(((s_m[lane] >> w) & 1u) != 0u);
        unsigned long long hit = 0ull;
        bool open = true;   // some pixel of this quadrant is still live
        while (bits && open) {
            const int j = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const uint32_t contributor = base + (uint32_t)j + 1u;
            Hit h;
            const bool valid = intersect(xl, yl, e0, e1, e2, e3, h) & (T > 0.f);
            if (ballot64(valid) == 0) continue;
            hit |= 1ull << j;
            const float4 e4 = s_e[4][j], e5 = s_e[5][j];
            if (valid) {
                const float test_T = T * (1.f - h.alpha);
                const bool go = !(test_T < kTStop);   // else: done, and this entry is NOT blended
                if (go) {
                    const float wgt = h.alpha * T;
                    const float A = 1.f - T;
                    const float mm = kFN * (1.f - kNear * fast_rcp(h.depth));
                    dist += (mm * mm * A + M2 - 2.f * mm * M1) * wgt;
                    Dsum += h.depth * wgt;
                    M1 += mm * wgt;
                    M2 += mm * mm * wgt;
                    if (T > 0.5f) { med = h.depth; medc = contributor; }
                    N0 += e4.x * wgt; N1 += e4.y * wgt; N2 += e4.z * wgt;
                    C0 += e4.w * wgt; C1 += e5.x * wgt; C2 += e5.y * wgt;
                    lastc = contributor;
                }
                T = go ? test_T : -T;
            }
            if (ballot64(T > 0.f) == 0) open = false;
        }
        s_hitq[w][lane] = (uint8_t)((hit >> lane) & 1ull);
        const bool quadrant_done = ballot64(T > 0.f) == 0ull;   // (evaluated by all 64 lanes, then acted on by one)
        if (lane == 0 && quadrant_done) s_alive[w] = 0u;
        __syncthreads();
        if (stager) {   // the round's hit masks: stored behind the NEXT staging (or behind the walk)
            hm_prev = (uint32_t)s_hitq[0][se] | ((uint32_t)s_hitq[1][se] << 1) | ((uint32_t)s_hitq[2][se] << 2) | ((uint32_t)s_hitq[3][se] << 3);
            n_prev = n; base_prev = base;
        }
        __syncthreads();   // (s_hitq / s_m / s_e are rewritten by the next round)
    }
    if (stager && hit_mask && (uint32_t)se < n_prev) hit_mask[range.x + base_prev + se] = (uint16_t)((hm_prev & 3u) | ((hm_prev >> 2) << 8));
    const size_t HW = (size_t)f.H * f.W;
    if (px < f.W && py < f.H) {
        const size_t pix = (size_t)py * f.W + px;
        const float Tq = fabsf(T);
        if (final_T) {
            final_T[pix] = Tq; final_T[HW + pix] = M1; final_T[2 * HW + pix] = M2;
            n_contrib[pix] = lastc; n_contrib[HW + pix] = medc;
        }
        out_color[pix] = C0 + Tq * f.bg[0];
        out_color[HW + pix] = C1 + Tq * f.bg[1];
        out_color[2 * HW + pix] = C2 + Tq * f.bg[2];
        out_allmap[pix] = Dsum;
        out_allmap[HW + pix] = 1.f - Tq;
        out_allmap[2 * HW + pix] = N0; out_allmap[3 * HW + pix] = N1; out_allmap[4 * HW + pix] = N2;
        out_allmap[5 * HW + pix] = med;
        out_allmap[6 * HW + pix] = dist;
    }
}

// The reference's 16x16 tile with three colour channels picks its mapping per FRAME, on the device: counts[0] = D, the frame's duplicates,
// counts[1] = its visible Gaussians (both fall out of the emission scan, radix_sort.hip).  Splats that touch few tiles touch few 4x4 cells
// of a quadrant, and the rows win; large splats fill quadrants, and one entry on all 64 lanes wins (3 M Gaussians: 1280x720, D / visible
// = 3.4: 0.620 vs 0.696 ms; 1920x1080, 5.2: 0.858 vs 0.885; 2560x1440, 7.4: 1.182 vs 1.164; 3840x2160, 13: 2.13 vs 1.94).  Both bodies
// produce the same bits, so the choice is invisible in the results.
constexpr uint32_t kRowsBelowDuplicatesPerVisibleX2 = 13;   // rows if D / visible < 6.5
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(6, 6)))
void render_forward_auto_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ point_list,
                                const float4* __restrict__ recs, float* __restrict__ out_color, float* __restrict__ out_allmap, float* __restrict__ final_T,
                                uint32_t* __restrict__ n_contrib, uint16_t* __restrict__ hit_mask, const uint32_t* __restrict__ counts) {
    __shared__ float4 s_e[entry_quads<3>()][kWave];
    __shared__ uint8_t s_hit[2][kWave];
    const unsigned long long D = counts[0], V = counts[1];
    if (2ull * D < (unsigned long long)kRowsBelowDuplicatesPerVisibleX2 * V)
        render_forward_rows_body<2, 1, 2>(s_e, s_hit, f, ranges, tile_order, point_list, recs, out_color, out_allmap, final_T, n_contrib, hit_mask);
    else
        render_forward_body<false, 3, 2, 1, 2>(s_e, f, ranges, tile_order, point_list, recs, nullptr, out_color, out_allmap, final_T, n_contrib, hit_mask, 1, nullptr);
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
                const unsigned long long vb = ballot64(valid), ub = ballot64(h.use3d);
                if (lane == 0) { valid_bits[(size_t)(range.x + base + j) * NQ + q] = vb; use3d_bits[(size_t)(range.x + base + j) * NQ + q] = ub; }
            }
        }
    }
}

// launchers ---------------------------------------------------------------------------------------
// Tile shapes (BASELINE config 5's sweep): the reference's 16x16 plus 8x8, 16x8, 32x8, 32x16 = QX x QY quadrants of 8x8
// pixels, i.e. 1 / 2 / 4 / 8 pixels per lane.  Every shape carries the 6- / 9-channel passes; only the reference shape the counter variant.
#define SR_FOR_TILE_SHAPE(F)                                                    \
    if (f.tile_w == 16 && f.tile_h == 16) { F(2, 2); }                          \
    else if (f.tile_w == 8 && f.tile_h == 8) { F(1, 1); }                       \
    else if (f.tile_w == 16 && f.tile_h == 8) { F(2, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 8) { F(4, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 16) { F(4, 2); }                     \
    else return hipErrorInvalidValue;

// (bit 5 of `flags`: the cooperative forward -- measured and NOT picked by itself: see render_forward_coop_kernel;
//  bit 6: the row-mapped kernel writing CELL-granular hit masks -- what render_backward_rows_kernel reads; 16x16, three channels, culling on)
// flags: bit 0 = quadrant culling on (SR_FLAG_NO_QUADRANT_CULL clear), bit 1 = counter variant (counters != NULL), bit 2 / bit 3 = the row-mapped /
// the quadrant-mapped kernel forced (else, for the 16x16 tile with three channels and culling on, the device picks per frame: frame_counts)
hipError_t launch_render_forward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                 const float* extra, float* out_color, float* out_allmap, float* final_T, uint32_t* n_contrib,
                                 uint16_t* hit_mask, int flags, unsigned long long* counters, const uint32_t* frame_counts, hipStream_t s) {
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
        else if (cull && !(flags & (4 | 8)) && (flags & 32))   // (explicitly asked for: SR_FLAG_COOP_BACKWARD in sr_forward_render)
                                hipLaunchKernelGGL(render_forward_coop_kernel, dim3(n_tiles), dim3(4 * kWave), 0, s, f, ranges, tile_order, point_list, recs, out_color,
                                                   out_allmap, final_T, n_contrib, hit_mask);
        else if (flags & 64)    hipLaunchKernelGGL((render_forward_rows_kernel<2, 1, 2, true>), dim3((n_tiles + kXcds - 1) / kXcds * kXcds * 2), block, 0, s, f, ranges,
                                                   tile_order, point_list, recs, out_color, out_allmap, final_T, n_contrib, hit_mask);   // (cell-granular hit masks)
        else if (flags & 4)     hipLaunchKernelGGL((render_forward_rows_kernel<2, 1, 2>), dim3((n_tiles + kXcds - 1) / kXcds * kXcds * 2), block, 0, s, f, ranges,
                                                   tile_order, point_list, recs, out_color, out_allmap, final_T, n_contrib, hit_mask);
        else if (cull && !(flags & 8) && frame_counts)
                                hipLaunchKernelGGL(render_forward_auto_kernel, dim3((n_tiles + kXcds - 1) / kXcds * kXcds * 2), block, 0, s, f, ranges,
                                                   tile_order, point_list, recs, out_color, out_allmap, final_T, n_contrib, hit_mask, frame_counts);
        else                    SR_LAUNCH_FWD(false, 3, 2, 1, 2);
    } else if (f.colors != 3 && f.tile_w == 32 && f.tile_h == 16) {
        // 32x16 with 6 / 9 channels: two 32x8 band waves per tile, like the 3-channel pass (K7 walks the list once per band there: render_bwd.hip)
        if (f.colors == 9) { SR_LAUNCH_FWD(false, 9, 4, 1, 2); } else { SR_LAUNCH_FWD(false, 6, 4, 1, 2); }
    } else if (f.colors != 3) {
        // the shared-geometry passes on the other shapes with up to four pixels per lane: 8x8, 16x8, 32x8
        if (f.tile_h != 8) return hipErrorInvalidValue;
#define SR_FWD_NC(QX)                                                                                        \
        { if (f.colors == 9) { SR_LAUNCH_FWD(false, 9, QX, 1, 1); } else { SR_LAUNCH_FWD(false, 6, QX, 1, 1); } }
        if (f.tile_w == 8) SR_FWD_NC(1) else if (f.tile_w == 16) SR_FWD_NC(2) else if (f.tile_w == 32) SR_FWD_NC(4) else return hipErrorInvalidValue;
#undef SR_FWD_NC
    } else {
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
