// render_bwd.hip -- K7, the per-tile backward blend (see render.hip for the forward and the algebra of the suffix-sum form).
// A translation unit of its own: it is compiled with the max-ILP machine scheduler (build.py), which interleaves the independent
// chains of a quadrant block better (K7 1.688 -> 1.670 ms) but costs the forward kernel a spill at its 80-register budget.
#include "blend_common.h"

#ifndef SR_K7_NINE_BANDED
#define SR_K7_NINE_BANDED 0
#endif

namespace sr {

// Stores the gradient record a lane holds in its row of s_out (the round that just ended) at the duplicate's emission index `slot`.
// kShiftInLds: the shift (ox, oy) of the moments waits in slots 22, 23 of the row (a 21-value record leaves them unused), else in registers.
// add: a banded walk's second pass adds to the record the first pass left (same wave, program order, a fence in between).
template <int kGQ, bool kShiftInLds>
__device__ __forceinline__ void flush_record(const float* __restrict__ row, float4* __restrict__ inst_grads, uint8_t* __restrict__ written, uint32_t slot,
                                             float ox, float oy, bool add) {
    const float4* accl = reinterpret_cast<const float4*>(row);
    float4 a0 = accl[0], a1 = accl[1], a2 = accl[2], a3 = accl[3], a4 = accl[4], a5 = accl[5], a6 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kGQ == 7) a6 = accl[6];
    if (kShiftInLds) { ox = a5.z; oy = a5.w; a5.z = 0.f; a5.w = 0.f; }
    a0.w = fmaf(ox, a0.x, a0.w); a1.x = fmaf(ox, a0.y, a1.x); a1.y = fmaf(ox, a0.z, a1.y);
    a1.z = fmaf(oy, a0.x, a1.z); a1.w = fmaf(oy, a0.y, a1.w); a2.x = fmaf(oy, a0.z, a2.x);
    float4* o = inst_grads + (size_t)slot * kGQ;
    if (add && written[slot]) {   // the upper band left a record for this duplicate: add to it
#define SR_ADD4(A, K) { const float4 p = o[K]; A.x += p.x; A.y += p.y; A.z += p.z; A.w += p.w; }
        SR_ADD4(a0, 0) SR_ADD4(a1, 1) SR_ADD4(a2, 2) SR_ADD4(a3, 3) SR_ADD4(a4, 4) SR_ADD4(a5, 5)
        if (kGQ == 7) SR_ADD4(a6, 6)
#undef SR_ADD4
    }
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;
    if (kGQ == 7) o[6] = a6;
    written[slot] = 1;   // K8 reads this 1-B flag (zeroed per call) before it touches the record
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
// Output: one 96-B gradient record per (tile, Gaussian) duplicate, stored at the duplicate's EMISSION index
// (first[gid] + its index inside the Gaussian's tile rectangle; first[] = FrameDev.first, fetched next to the record), where the
// records of one Gaussian are contiguous; K8 sums them.  Only entries with a contributing pixel get a record, and a 1 in
// `written[]` at the same index (zeroed per call).  Gradient record slots: see common.h.
// Register budget: three waves per SIMD (<= 168 VGPRs) wherever the per-pixel state allows it -- the loop is latency-bound
// at two -- i.e. up to four pixels per lane with three colour channels.
// BANDS = 2 (32x16 tile with 6 / 9 colour channels: eight pixels per lane do not fit the register file there): the wave walks the list
// TWICE, once per 32x8 band (QY = 1 quadrant row each, four pixels per lane), and the second walk ADDS its sums to the records the first
// one wrote -- same wave, program order, a fence in between -- so a record still holds the whole tile's contribution and K8 is unchanged.
// kXG = false (6 / 9 channels, SR_FLAG_NO_PRECOMP_COLOR_GRAD): nobody wants dL/dcolors_precomp -- the one-hot class channels of
// render_semantic / render_and_semantic are constants -- so their six per-entry sums (and, with 9 channels, the second wave reduction
// they need) are not formed; the channels still feed dL/dalpha.  The record keeps its size, the slots stay zero.
template <int NC, int QX, int QY, int BANDS = 1, bool kXG = true>
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
    __shared__ float4 s_zero[6][kZeroCopies];   // the per-entry accumulators start from zeros read from the LDS (blend_common.h lds_zeros_load)
    const int lane = threadIdx.x;
    lds_zeros_init<6>(s_zero, lane);
    const int tile = (int)tile_order[blockIdx.x];   // longest lists first
    constexpr int NQ = QX * QY;   // 8x8 quadrants per tile = pixels per lane
    static_assert(BANDS == 1 || QY == 1, "a banded walk handles one quadrant row per band");
    const int tx0 = (tile % f.tiles_x) * (QX * 8), tile_y0 = (tile / f.tiles_x) * (QY * 8 * BANDS);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)(tile_y0 + QY * 4 * BANDS);   // the TILE's centre: the local origin of staging and moments
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = ranges[tile];
    const uint32_t count = range.y - range.x;
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];
  for (int band = 0; band < BANDS; ++band) {
    const int ty0 = tile_y0 + band * (QY * 8);
    if (BANDS > 1 && band > 0) __threadfence();   // the first walk's records and `written` flags, visible to this wave's loads

    // per-pixel constants (upstream gradients folded with the forward's final accumulators) and state
    const float xl0 = (float)(lx - QX * 4), yl0 = (float)(ly - QY * 4 * BANDS + band * (QY * 8));   // tile-local pixel of quadrant 0; quadrant q adds 8*(q%QX, q/QX)
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
    //
    // Memory pipeline of the walk (round 6).  Nothing that comes back from memory is touched in the round that asks for it:
    //   * the list entry (gid) of round r - 2 is requested during round r, the record of round r - 1 (address = that gid, which arrived a
    //     round ago) too, RAW -- first[] and first_base[] stay two registers and the hit mask stays undecoded until round r - 1 stages them;
    //   * the records of round r are STORED at the top of round r - 1, behind its staging: the wave's vector-memory counter retires in
    //     issue order, so the wait for round r - 1's record loads -- issued before those stores -- never waits for a store.
    // Before: `gid = point_list[pos]; load_record(recs, gid, ...)` and `first[gid] + first_base[..]` inside the prefetch made the wave wait
    // for two dependent memory round trips (and for the stores of the previous flush in front of them) in EVERY round -- SQ_WAIT_INST_ANY was
    // 26 % of the wave cycles (profiles/r05_c3_sq_counters.json).
    const int rounds = (int)((total + kWave - 1) / kWave);
    float4 nr[kRecQuads], nx = make_float4(0.f, 0.f, 0.f, 0.f), ny = nx;
    uint32_t nfirst = 0, nfbase = 0, nhraw = 0, gid_ahead = 0;
    if (rounds > 0 && (uint32_t)((rounds - 1) * kWave + lane) < total) {
        const uint32_t pos = range.x + (rounds - 1) * kWave + lane;
        const uint32_t gid = point_list[pos];
        load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
        if (NC == 6) nx = load_extra(extra, gid, 3);
        if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
        nhraw = hit_mask[pos];
    }
    if (rounds > 1) gid_ahead = point_list[range.x + (rounds - 2) * kWave + lane];   // (every round but the last one is full)
    // (with 21 values the lanes of value slots 21..23 hold copies of other totals: they must not reach s_out, whose slots 22, 23 are in use)
    const bool holds_total = reduce24_holds_total(lane) && (NC != 3 || reduce24_index(lane) < 21);
    bool pend = false;             // this lane holds a record of the previous round that is not stored yet
    uint32_t pslot = 0;
    // the shift of the moments to the Gaussian's own centre (see the staging below) waits for the flush in the lane's own row of s_out, in
    // the two slots a 21-value record leaves unused (three channels: no registers across the entry loop); in registers otherwise
    constexpr bool kShiftInLds = NC == 3;
    float pox = 0.f, poy = 0.f;
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        uint32_t m = 0, slot = 0;
        float ox = 0.f, oy = 0.f;
        wait_vector_memory();   // (loads and stores of the previous round: see blend_common.h)
        if ((uint32_t)lane < n) {
            (void)stage_entry<QX, QY, NC>(nr, nx, ny, Xc, Yc, 0, s_e, lane);
            // (entry, quadrant) pairs that reached a pixel in the forward: exact, no culling test needed here
            m = (decode_hits<QX, QY * BANDS>((uint16_t)nhraw) >> (band * QX * QY)) & ((1u << (QX * QY)) - 1u);
            slot = emission_index(nr, nfirst + nfbase, tile % f.tiles_x, tile / f.tiles_x, f);
            uint32_t need = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) need |= (rbase + lane < quad_last[q]) ? (1u << q) : 0u;
            m &= need;
            // Sx, Sy from tile-local coordinates to coordinates relative to the Gaussian's OWN centre (cx, cy): sum (xl - mx) dp = sum xl dp -
            // mx S0 with mx = cx - Xc.  K8 sums these over the Gaussian's tiles and works with Tu - cx Tw, Tv - cy Tw: the same dL/dT as
            // with moments about the image origin, without the cancellation of pixel coordinates ~1000 against extents of a few pixels
            // (clamped into the image: the moments of a splat whose centre projects far off-screen are taken about the nearest image point)
            const float mx = nr[2].y - Xc, my = nr[2].z - Yc;   // (the staged centre: same expression, same bits as stage_entry's)
            ox = -fminf(fmaxf(mx, -Xc), (float)(f.W - 1) - Xc); oy = -fminf(fmaxf(my, -Yc), (float)(f.H - 1) - Yc);
        }
        // (the prefetch comes BEFORE the stores of the flush: the wait for `gid_ahead` -- the last load of the previous round -- then has
        // the same number of younger memory operations behind it on every path, none, instead of "seven stores or none")
        if (rd > 0) {  // next round is always full
            const uint32_t gid = gid_ahead;
            load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
            if (NC == 6) nx = load_extra(extra, gid, 3);
            if (NC == 9) { nx = load_extra(extra, gid, 0); ny = load_extra(extra, gid, 3); }
            nhraw = hit_mask[range.x + rbase - kWave + lane];
            if (rd > 1) gid_ahead = point_list[range.x + rbase - 2 * kWave + lane];
        }
        if (pend) flush_record<kGQ, kShiftInLds>(&s_out[lane][0], inst_grads, written, pslot, pox, poy, BANDS > 1 && band > 0);   // one 96-B store per lane whose entry of the previous round got a contribution
        {
            float4* z = reinterpret_cast<float4*>(&s_out[lane][0]);
#pragma unroll
            for (int k = 0; k < kGQ; ++k) z[k] = (kShiftInLds && k == 5) ? make_float4(0.f, 0.f, ox, oy) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        unsigned long long bits = ballot64(m != 0);
        // entries of this round that get a record: those with an (entry, quadrant) pair that reached a pixel in the forward.  (Such a
        // pair has a valid lane here too -- same decisions, bit for bit -- unless every pixel it reached stopped at the transmittance
        // floor instead of blending: that rare entry gets a record of zeros rather than a wave-wide `any lane valid` test per pair.)
        const unsigned long long wrote = bits;
        while (bits) {
            const int j = 63 - __clzll((long long)bits);
            bits &= ~(1ull << j);
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            // Three channels: normal and colour of the entry are fetched HERE, beside the geometry, once per entry -- inside the `valid` block (once per
            // quadrant test, as the 6- / 9-channel instantiations at their register limit still do) every test waited for its own LDS round trip.
            // 162 -> 168 registers: exactly the three-waves budget.  K7 1.605 -> 1.560 ms at C3, bit-identical (same-box A/B).
            constexpr bool kEntryColoursUpFront = NC == 3 || QX * QY >= 4;   // (6 / 9 channels on four or more pixels per lane run two waves per SIMD: registers to spare)
            float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f), e5 = e4;
            if (kEntryColoursUpFront) { e4 = s_e[4][j]; e5 = s_e[5][j]; }
            float4 e6_up = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NC == 9 && kEntryColoursUpFront) e6_up = s_e[6][j];
            const uint32_t cidx = rbase + (uint32_t)j;  // 0-based contributor index
            constexpr int NV = (NC == 3 || !kXG) ? 21 : 24;   // slots 21..23 carry colour channels 3..5 only
            float v[24];
            lds_zeros_load<NV>(s_zero, j, v);   // (no v_mov: six broadcast LDS loads beside the entry's own)
            float w6 = 0.f, w7 = 0.f, w8 = 0.f;   // colour channels 6..8 (9-channel variant)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(mj & (1u << q))) continue;  // wave-uniform
                Hit h;
                const float xq = xl0 + (float)((q % QX) * 8), yq = yl0 + (float)((q / QX) * 8);
                const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (cidx < lastc[q]);
                if (valid) {
                    if (!kEntryColoursUpFront) { e4 = s_e[4][j]; e5 = s_e[5][j]; }
                    const float Twx = e2.y, Twy = e2.z;
                    const float one_m_inv = fast_rcp(1.f - h.alpha);
                    T[q] *= one_m_inv;                 // transmittance in front of this entry
                    const float w = h.alpha * T[q];
                    // psi = rgb.g + depth g_depth + n.gn + (a2 + m (m a0 - 2 a1)), m = the depth metric; dL/dz = w (2 (m a0 - a1) dm/dz + g_depth),
                    // dm/dz = kFN kNear / depth^2.  t1 = m a0 - a1 serves both: 8 instructions where the literal transcription took 11.  (The
                    // three distortion terms cancel to the variance of m along the ray: they are combined in ONE fma before anything else is
                    // added -- seeding the colour chain with a2 saves another instruction and costs a digit under a distortion-weighted loss.)
                    float phi = fmaf(e4.w, gr[q], fmaf(e5.x, gg[q], fmaf(e5.y, gb[q], fmaf(h.depth, g_depth[q],
                                fmaf(e4.x, gn0[q], fmaf(e4.y, gn1[q], e4.z * gn2[q]))))));
                    if (NC >= 6) phi = fmaf(e5.z, gc3[q], fmaf(e5.w, gc4[q], fmaf(e3.w, gc5[q], phi)));
                    if (NC == 9) { const float4 e6 = kEntryColoursUpFront ? e6_up : s_e[6][j]; phi = fmaf(e6.x, gc6[q], fmaf(e6.y, gc7[q], fmaf(e6.z, gc8[q], phi))); }
                    const float inv_depth = fast_rcp(h.depth);
                    const float m_d = fmaf(inv_depth, -kFN * kNear, kFN);
                    const float t1 = fmaf(m_d, a0[q], -a1[q]);
#if SR_DETACH_WEIGHT
                    const float psi = phi;   // upstream DETACH_WEIGHT: the distortion does not differentiate through the blend weights
#else
                    const float psi = phi + fmaf(m_d, t1 - a1[q], a2[q]);
#endif
                    const float dL_dalpha = T[q] * psi - one_m_inv * Z[q];
                    Z[q] = fmaf(w, psi, Z[q]);
                    const float med_add = (cidx == medc[q] - (SR_MEDIAN_CONTRIBUTOR_MINUS_ONE ? 1u : 0u)) ? g_median[q] : 0.f;
                    const float dL_dz = fmaf(w, fmaf(t1 * (inv_depth * inv_depth), 2.f * kFN * kNear, g_depth[q]), med_add);
                    const float dL_dG = e3.z * dL_dalpha;
                    if (NC != 6 || kXG) { v[18] += w * gr[q]; v[19] += w * gg[q]; v[20] += w * gb[q]; }   // (6 channels: all of them precomputed)
                    if (NC >= 6 && kXG) { v[21] += w * gc3[q]; v[22] += w * gc4[q]; v[23] += w * gc5[q]; }
                    if (NC == 9 && kXG) { w6 += w * gc6[q]; w7 += w * gc7[q]; w8 += w * gc8[q]; }
                    v[15] += w * gn0[q]; v[16] += w * gn1[q]; v[17] += w * gn2[q];
                    v[14] += h.G * dL_dalpha;
                    v[11] += dL_dz;   // (both paths)
                    if (h.use3d) {
                        const float gG = -dL_dG * h.G;
                        const float dpx = (gG * h.sx + dL_dz * Twx) * h.pz_inv, dpy = (gG * h.sy + dL_dz * Twy) * h.pz_inv;
                        const float dpz = -(dpx * h.sx + dpy * h.sy);
                        // moments of dL/dp in tile-local pixel coordinates (shifted to global ones when the record is written);
                        // the cross products happen once per Gaussian in K8
                        v[0] += dpx; v[1] += dpy; v[2] += dpz;
                        v[3] = fmaf(xq, dpx, v[3]); v[4] = fmaf(xq, dpy, v[4]); v[5] = fmaf(xq, dpz, v[5]);
                        v[6] = fmaf(yq, dpx, v[6]); v[7] = fmaf(yq, dpy, v[7]); v[8] = fmaf(yq, dpz, v[8]);
                        v[9] = fmaf(dL_dz, h.sx, v[9]); v[10] = fmaf(dL_dz, h.sy, v[10]);
                    } else {
                        const float gG = -dL_dG * h.G * kFilterInvSquare;
                        v[12] = fmaf(gG, h.dx, v[12]);
                        v[13] = fmaf(gG, h.dy, v[13]);
                    }
                }
            }
            {
                const float tot = wave_reduce24<NV>(v, lane);
                if (holds_total) s_out[j][reduce24_index(lane)] = tot;
                if (NC == 9 && kXG) {
                    const float t3 = wave_reduce3(w6, w7, w8);   // row 0: channel 6, row 1: channel 8, row 2: channel 7
                    if ((lane & 15) == 0 && lane < 48) s_out[j][24 + (lane == 0 ? 0 : (lane == 16 ? 2 : 1))] = t3;
                }
            }
        }
        pend = ((wrote >> lane) & 1ull) != 0ull; pslot = slot;
        if (!kShiftInLds) { pox = ox; poy = oy; }
    }
    if (pend) flush_record<kGQ, kShiftInLds>(&s_out[lane][0], inst_grads, written, pslot, pox, poy, BANDS > 1 && band > 0);
  }   // band
}


// ---------------------------------------------------------------------------------------------
// K7, cooperative form (round 6): FOUR waves per 16x16 tile, one per 8x8 quadrant (one pixel per lane), one workgroup.
// For frames with FEW tiles -- the reference's own runs render 480x320 = 600 tiles [REF /root/reference/README.md:195-207] -- where one wave
// per tile leaves most of the GPU's 3 072 wave slots empty.  Same tile lists, same staging (each wave stages 16 of the round's 64 entries
// into the SHARED s_e), same per-pair arithmetic; every wave walks the entries whose hit mask carries ITS quadrant, reduces its own 21
// sums per entry into its slice of s_part, and after a barrier the workgroup adds the (up to four) slices of every entry in quadrant
// order and stores ONE record per duplicate, as K7 does -- K8 is unchanged, no atomics, deterministic.  What it costs: a wave reduction per
// (entry, quadrant) pair instead of per entry (7.66 M instead of 4.07 M at C3: that is why it is NOT the kernel for full-size frames --
// DESIGN.md 4, measured in tools/notes_round6_measured.md) and three barriers per round.  Gradients equal K7's up to the order of a four-term sum.
// ---------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(4 * kWave) __attribute__((amdgpu_waves_per_eu(4, 4)))
void render_backward_coop_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                 const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
                                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dallmap,
                                 const uint16_t* __restrict__ hit_mask, float4* __restrict__ inst_grads, uint8_t* __restrict__ written) {
    static_assert(NC == 3, "three colour channels (the operator's own call); the 6- / 9-channel passes keep the one-wave kernel");
    constexpr int kGQ = kGradQuads, kStage = kWave / 4;   // entries of a round staged per wave
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    __shared__ uint32_t s_m[kWave], s_slot[kWave], s_ql[4];
    __shared__ float s_ox[kWave], s_oy[kWave];
    __shared__ __attribute__((aligned(16))) float s_part[4][kWave][kGQ * 4];
    __shared__ float4 s_zero[6][kZeroCopies];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (w == 0) lds_zeros_init<6>(s_zero, lane);   // (read behind the first __syncthreads below)
    const int tile = (int)tile_order[blockIdx.x];
    const int tx0 = (tile % f.tiles_x) * 16, ty0 = (tile / f.tiles_x) * 16;
    const float Xc = (float)(tx0 + 8), Yc = (float)(ty0 + 8);
    const int lx = lane & 7, ly = lane >> 3, qx = w & 1, qy = w >> 1;   // wave w = quadrant (w % 2, w / 2) = bit w of the hit masks
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)f.H * f.W;
    const int px = tx0 + qx * 8 + lx, py = ty0 + qy * 8 + ly;
    const bool inside = px < f.W && py < f.H;
    const size_t pix = inside ? (size_t)py * f.W + px : 0;
    const float xq = (float)(qx * 8 + lx - 8), yq = (float)(qy * 8 + ly - 8);
    const float T_final = inside ? final_T[pix] : 0.f;
    const float fin_D = inside ? final_T[HW + pix] : 0.f, fin_D2 = inside ? final_T[2 * HW + pix] : 0.f;
    const uint32_t lastc = inside ? n_contrib[pix] : 0u, medc = inside ? n_contrib[HW + pix] : 0u;
    const float gr = inside ? dL_dcolor[pix] : 0.f, gg = inside ? dL_dcolor[HW + pix] : 0.f, gb = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float g_depth = inside ? dL_dallmap[pix] : 0.f, g_accum = inside ? dL_dallmap[HW + pix] : 0.f;
    const float gn0 = inside ? dL_dallmap[2 * HW + pix] : 0.f, gn1 = inside ? dL_dallmap[3 * HW + pix] : 0.f, gn2 = inside ? dL_dallmap[4 * HW + pix] : 0.f;
    const float g_median = inside ? dL_dallmap[5 * HW + pix] : 0.f, g_reg = inside ? dL_dallmap[6 * HW + pix] : 0.f;
    const float a0 = (1.f - T_final) * g_reg, a1 = fin_D * g_reg, a2 = fin_D2 * g_reg;
    float T = T_final, Z = -T_final * (g_accum - (f.bg[0] * gr + f.bg[1] * gg + f.bg[2] * gb));
    {
        const uint32_t ql = wave_max_u32(lastc);   // deepest entry any pixel of this wave's quadrant needs
        if (lane == 0) s_ql[w] = ql;
    }
    __syncthreads();
    const uint32_t ql0 = s_ql[0], ql1 = s_ql[1], ql2 = s_ql[2], ql3 = s_ql[3];
    const uint32_t total = max(max(ql0, ql1), max(ql2, ql3));
    const int rounds = (int)((total + kWave - 1) / kWave);
    const bool stager = lane < kStage;
    const int se = w * kStage + lane;          // the entry of the round this lane stages (lanes < 16 of each wave)
    const bool holds_total = reduce24_holds_total(lane) && reduce24_index(lane) < 21;
    // the walk's memory pipeline as in the one-wave kernel: list entries two rounds ahead, records one, raw
    float4 nr[kRecQuads];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nfirst = 0, nfbase = 0, nhraw = 0, gid_ahead = 0;
    if (stager && rounds > 0 && (uint32_t)((rounds - 1) * kWave + se) < total) {
        const uint32_t pos = range.x + (rounds - 1) * kWave + se;
        const uint32_t gid = point_list[pos];
        load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
        nhraw = hit_mask[pos];
    }
    if (stager && rounds > 1) gid_ahead = point_list[range.x + (rounds - 2) * kWave + se];
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        wait_vector_memory();
        if (stager) {
            uint32_t m = 0;
            if ((uint32_t)se < n) {
                (void)stage_entry<2, 2, NC>(nr, zero4, zero4, Xc, Yc, 0, s_e, se);
                const uint32_t at = rbase + (uint32_t)se;
                const uint32_t need = (at < ql0 ? 1u : 0u) | (at < ql1 ? 2u : 0u) | (at < ql2 ? 4u : 0u) | (at < ql3 ? 8u : 0u);
                m = decode_hits<2, 2>((uint16_t)nhraw) & need;
                s_slot[se] = emission_index(nr, nfirst + nfbase, tile % f.tiles_x, tile / f.tiles_x, f);
                const float mx = nr[2].y - Xc, my = nr[2].z - Yc;
                s_ox[se] = -fminf(fmaxf(mx, -Xc), (float)(f.W - 1) - Xc); s_oy[se] = -fminf(fmaxf(my, -Yc), (float)(f.H - 1) - Yc);
            }
            s_m[se] = m;
            if (rd > 0) {   // (the next round is always full)
                const uint32_t gid = gid_ahead;
                load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
                nhraw = hit_mask[range.x + rbase - kWave + se];
                if (rd > 1) gid_ahead = point_list[range.x + rbase - 2 * kWave + se];
            }
        }
        __syncthreads();
        const uint32_t mine = s_m[lane];
        unsigned long long bits = ballot64(((mine >> w) & 1u) != 0u);   // the round's entries that reached a pixel of THIS quadrant in the forward
        while (bits) {
            const int j = 63 - __clzll((long long)bits);
            bits &= ~(1ull << j);
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const float4 e4 = s_e[4][j], e5 = s_e[5][j];   // (up front, beside the geometry: see render_backward_kernel)
            const uint32_t cidx = rbase + (uint32_t)j;
            float v[24];
            lds_zeros_load<21>(s_zero, j, v);
            Hit h;
            const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (cidx < lastc);
            if (valid) {   // (the per-pair arithmetic of render_backward_kernel, one quadrant)
                const float Twx = e2.y, Twy = e2.z;
                const float one_m_inv = fast_rcp(1.f - h.alpha);
                T *= one_m_inv;
                const float wgt = h.alpha * T;
                const float phi = fmaf(e4.w, gr, fmaf(e5.x, gg, fmaf(e5.y, gb, fmaf(h.depth, g_depth, fmaf(e4.x, gn0, fmaf(e4.y, gn1, e4.z * gn2))))));
                const float inv_depth = fast_rcp(h.depth);
                const float m_d = fmaf(inv_depth, -kFN * kNear, kFN);
                const float t1 = fmaf(m_d, a0, -a1);
#if SR_DETACH_WEIGHT
                const float psi = phi;
#else
                const float psi = phi + fmaf(m_d, t1 - a1, a2);
#endif
                const float dL_dalpha = T * psi - one_m_inv * Z;
                Z = fmaf(wgt, psi, Z);
                const float med_add = (cidx == medc - (SR_MEDIAN_CONTRIBUTOR_MINUS_ONE ? 1u : 0u)) ? g_median : 0.f;
                const float dL_dz = fmaf(wgt, fmaf(t1 * (inv_depth * inv_depth), 2.f * kFN * kNear, g_depth), med_add);
                const float dL_dG = e3.z * dL_dalpha;
                v[18] = wgt * gr; v[19] = wgt * gg; v[20] = wgt * gb;
                v[15] = wgt * gn0; v[16] = wgt * gn1; v[17] = wgt * gn2;
                v[14] = h.G * dL_dalpha;
                v[11] = dL_dz;
                if (h.use3d) {
                    const float gG = -dL_dG * h.G;
                    const float dpx = (gG * h.sx + dL_dz * Twx) * h.pz_inv, dpy = (gG * h.sy + dL_dz * Twy) * h.pz_inv;
                    const float dpz = -(dpx * h.sx + dpy * h.sy);
                    v[0] = dpx; v[1] = dpy; v[2] = dpz;
                    v[3] = xq * dpx; v[4] = xq * dpy; v[5] = xq * dpz;
                    v[6] = yq * dpx; v[7] = yq * dpy; v[8] = yq * dpz;
                    v[9] = dL_dz * h.sx; v[10] = dL_dz * h.sy;
                } else {
                    const float gG = -dL_dG * h.G * kFilterInvSquare;
                    v[12] = gG * h.dx; v[13] = gG * h.dy;
                }
            }
            const float tot = wave_reduce24<21>(v, lane);
            if (holds_total) s_part[w][j][reduce24_index(lane)] = tot;
        }
        __syncthreads();
        {   // one record per entry with a hit: the quadrants' slices added in quadrant order; thread (entry j, group g) takes quads g and g + 4
            const int j = lane, g = w;
            const uint32_t m = s_m[j];
            if (m) {
                auto sum_quad = [&](int k) {
                    float4 a = zero4;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (m & (1u << q)) { const float4 p = *reinterpret_cast<const float4*>(&s_part[q][j][4 * k]); a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
                    return a;
                };
                const float ox = s_ox[j], oy = s_oy[j];
                float4* o = inst_grads + (size_t)s_slot[j] * kGQ;
                if (g == 0) {
                    float4 a0q = sum_quad(0), a4q = sum_quad(4);
                    a0q.w = fmaf(ox, a0q.x, a0q.w);
                    o[0] = a0q; o[4] = a4q;
                    written[s_slot[j]] = 1;   // K8 reads this 1-B flag (zeroed per call) before it touches the record
                } else if (g == 1) {
                    const float4 s0 = sum_quad(0);
                    float4 a1q = sum_quad(1), a5q = sum_quad(5);
                    a1q.x = fmaf(ox, s0.y, a1q.x); a1q.y = fmaf(ox, s0.z, a1q.y); a1q.z = fmaf(oy, s0.x, a1q.z); a1q.w = fmaf(oy, s0.y, a1q.w);
                    a5q.y = 0.f; a5q.z = 0.f; a5q.w = 0.f;   // slots 21..23: not in use with three channels (the reduction leaves copies there)
                    o[1] = a1q; o[5] = a5q;
                } else if (g == 2) {
                    const float4 s0 = sum_quad(0);
                    float4 a2q = sum_quad(2);
                    a2q.x = fmaf(oy, s0.z, a2q.x);
                    o[2] = a2q;
                } else {
                    o[3] = sum_quad(3);
                }
            }
        }
        __syncthreads();   // (the next round's staging overwrites s_e / s_m / s_slot)
    }
}


// ---------------------------------------------------------------------------------------------
// K7, row-mapped (round 6; three colour channels, the 16x16 tile): the wave's four 16-lane rows are the four 4x4 CELLS of a quadrant and every
// row walks ITS OWN list -- the entries of the round that reached a pixel of its cell in the forward (render_forward_rows_kernel<.., kCells>
// writes the exact (entry, cell) bits) -- so one wave step serves four different entries instead of one entry on 64 lanes of which 0.29 take
// part (C3: 5.6 M steps instead of 7.66 M quadrant tests).  What it costs: the 21 sums are reduced per STEP inside each 16-lane row (two
// bank-masked DPP levels + two quad levels) and ADDED to the entry's row of s_out with LDS float atomics (one wave: program order, so the
// sums are deterministic), instead of once per entry across the wave.  Same staging, same `intersect`, same per-pair arithmetic, same
// records and flush as render_backward_kernel; the per-entry sums differ from its by the order of the additions.
// MEASURED at C3 (profiles/r06_c3rows_*): 2.55 ms against the one-wave kernel's 1.65 -- 5.56 M steps instead of 7.66 M tests, but (i) a step costs
// 182 vector instructions (per-lane entry addresses, 64-bit per-lane list cursors, the in-row reduction per step): SQ_INSTS_VALU 1 048 M against
// 1 008 M, nothing saved; (ii) the scatter-accumulate is what bounds it: a ds_add_f32 wave instruction with 16 active lanes on distinct addresses
// holds the CU's LDS pipe ~100 cycles (SQ_LDS_IDX_ACTIVE 1 274 M against 161 M: the LDS is busy 2.07 of the 2.6 ms, SQ_WAIT_INST_LDS 1 002 M of
// 3 930 M wave cycles).  With the quad levels of the reduction left to the LDS as well (six atomic adds per lane, four lanes per address): 9.8 ms.
// Opt-in (SR_FLAG_ROW_BACKWARD), tested, NOT a default anywhere: the measured answer to "walk per-cell contributor lists with LDS adds".
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 3)))
void render_backward_rows_kernel(FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                 const uint32_t* __restrict__ point_list, const float4* __restrict__ recs,
                                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dallmap,
                                 const uint16_t* __restrict__ hit_mask, float4* __restrict__ inst_grads, uint8_t* __restrict__ written) {
    constexpr int NC = 3, QX = 2, QY = 2, NQ = 4, kGQ = kGradQuads;
    __shared__ float4 s_e[entry_quads<NC>()][kWave];
    __shared__ __attribute__((aligned(16))) float s_out[kWave][kGQ * 4];
    const int lane = threadIdx.x;
    const int tile = (int)tile_order[blockIdx.x];
    const int tx0 = (tile % f.tiles_x) * 16, ty0 = (tile / f.tiles_x) * 16;
    const float Xc = (float)(tx0 + 8), Yc = (float)(ty0 + 8);
    // (its accumulators are cleared with v_mov: every row reads a different entry per step, the block of zeros has no uniform slot to follow)
    // row r = lane / 16 <-> cell (r & 1, r >> 1) of a quadrant; lane % 16 <-> pixel (l & 3, (l >> 2) & 3) of the cell (as in the row-mapped K6)
    const int row = lane >> 4;
    const int lx = (row & 1) * 4 + (lane & 3), ly = (row >> 1) * 4 + ((lane >> 2) & 3);
    const uint2 range = ranges[tile];
    const size_t HW = (size_t)f.H * f.W;
    const float bg0 = f.bg[0], bg1 = f.bg[1], bg2 = f.bg[2];
    const float xl0 = (float)(lx - 8), yl0 = (float)(ly - 8);
    float gr[NQ], gg[NQ], gb[NQ], gn0[NQ], gn1[NQ], gn2[NQ], g_depth[NQ], g_median[NQ], a0[NQ], a1[NQ], a2[NQ];
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
        const float bg_dot = bg0 * gr[q] + bg1 * gg[q] + bg2 * gb[q];
        a0[q] = (1.f - T_final) * g_reg; a1[q] = fin_D * g_reg; a2[q] = fin_D2 * g_reg;
        T[q] = T_final; Z[q] = -T_final * (g_accum - bg_dot);
        quad_last[q] = wave_max_u32(lastc[q]);
        total = max(total, quad_last[q]);
    }
    const int rounds = (int)((total + kWave - 1) / kWave);
    float4 nr[kRecQuads];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t nfirst = 0, nfbase = 0, nhraw = 0, gid_ahead = 0;
    if (rounds > 0 && (uint32_t)((rounds - 1) * kWave + lane) < total) {
        const uint32_t pos = range.x + (rounds - 1) * kWave + lane;
        const uint32_t gid = point_list[pos];
        load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
        nhraw = hit_mask[pos];
    }
    if (rounds > 1) gid_ahead = point_list[range.x + (rounds - 2) * kWave + lane];
    // after the in-row reduction the lanes of quad (bit 2, bit 3 of the lane) hold the row's totals of values k + 6 bit2 + 12 bit3, k = 0..5
    const int vbase = 6 * ((lane >> 2) & 1) + 12 * ((lane >> 3) & 1);
    bool pend = false;
    uint32_t pslot = 0;
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        uint32_t cm = 0, slot = 0;   // cm: bit 4 q + c = this lane's ENTRY reached a pixel of cell c of quadrant q in the forward
        float ox = 0.f, oy = 0.f;
        wait_vector_memory();
        if ((uint32_t)lane < n) {
            (void)stage_entry<QX, QY, NC>(nr, zero4, zero4, Xc, Yc, 0, s_e, lane);
            slot = emission_index(nr, nfirst + nfbase, tile % f.tiles_x, tile / f.tiles_x, f);
            uint32_t need = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) need |= (rbase + lane < quad_last[q]) ? (15u << (4 * q)) : 0u;
            cm = nhraw & need;
            const float mx = nr[2].y - Xc, my = nr[2].z - Yc;
            ox = -fminf(fmaxf(mx, -Xc), (float)(f.W - 1) - Xc); oy = -fminf(fmaxf(my, -Yc), (float)(f.H - 1) - Yc);
        }
        if (rd > 0) {
            const uint32_t gid = gid_ahead;
            load_record(recs, gid, nr); nfirst = f.first[gid]; nfbase = f.first_base[gid / kScanTile];
            nhraw = hit_mask[range.x + rbase - kWave + lane];
            if (rd > 1) gid_ahead = point_list[range.x + rbase - 2 * kWave + lane];
        }
        if (pend) flush_record<kGQ, true>(&s_out[lane][0], inst_grads, written, pslot, 0.f, 0.f, false);
        {
            float4* z = reinterpret_cast<float4*>(&s_out[lane][0]);
#pragma unroll
            for (int k = 0; k < kGQ; ++k) z[k] = (k == 5) ? make_float4(0.f, 0.f, ox, oy) : zero4;
        }
        const unsigned long long wrote = ballot64(cm != 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const unsigned long long b0 = ballot64(((cm >> (4 * q)) & 1u) != 0u), b1 = ballot64(((cm >> (4 * q + 1)) & 1u) != 0u),
                                     b2 = ballot64(((cm >> (4 * q + 2)) & 1u) != 0u), b3 = ballot64(((cm >> (4 * q + 3)) & 1u) != 0u);
            if ((b0 | b1 | b2 | b3) == 0ull) continue;   // wave-uniform
            unsigned long long brow = row == 0 ? b0 : (row == 1 ? b1 : (row == 2 ? b2 : b3));
            const float xq = xl0 + (float)((q % QX) * 8), yq = yl0 + (float)((q / QX) * 8);
            while (ballot64(brow != 0ull) != 0ull) {
                const bool act = brow != 0ull;
                const uint32_t j = act ? (uint32_t)(63 - __clzll((long long)brow)) : 0u;   // back to front
                brow &= ~(1ull << j);
                const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
                const uint32_t cidx = rbase + j;
                float v[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) {
                    v[k] = 0.f;
                    if (k < 21) asm volatile("" : "+v"(v[k]));
                }
                Hit h;
                const bool valid = intersect(xq, yq, e0, e1, e2, e3, h) & (cidx < lastc[q]) & act;
                if (valid) {   // (the per-pair arithmetic of render_backward_kernel)
                    const float4 e4 = s_e[4][j], e5 = s_e[5][j];
                    const float Twx = e2.y, Twy = e2.z;
                    const float one_m_inv = fast_rcp(1.f - h.alpha);
                    T[q] *= one_m_inv;
                    const float w = h.alpha * T[q];
                    const float phi = fmaf(e4.w, gr[q], fmaf(e5.x, gg[q], fmaf(e5.y, gb[q], fmaf(h.depth, g_depth[q],
                                      fmaf(e4.x, gn0[q], fmaf(e4.y, gn1[q], e4.z * gn2[q]))))));
                    const float inv_depth = fast_rcp(h.depth);
                    const float m_d = fmaf(inv_depth, -kFN * kNear, kFN);
                    const float t1 = fmaf(m_d, a0[q], -a1[q]);
#if SR_DETACH_WEIGHT
                    const float psi = phi;
#else
                    const float psi = phi + fmaf(m_d, t1 - a1[q], a2[q]);
#endif
                    const float dL_dalpha = T[q] * psi - one_m_inv * Z[q];
                    Z[q] = fmaf(w, psi, Z[q]);
                    const float med_add = (cidx == medc[q] - (SR_MEDIAN_CONTRIBUTOR_MINUS_ONE ? 1u : 0u)) ? g_median[q] : 0.f;
                    const float dL_dz = fmaf(w, fmaf(t1 * (inv_depth * inv_depth), 2.f * kFN * kNear, g_depth[q]), med_add);
                    const float dL_dG = e3.z * dL_dalpha;
                    v[18] += w * gr[q]; v[19] += w * gg[q]; v[20] += w * gb[q];
                    v[15] += w * gn0[q]; v[16] += w * gn1[q]; v[17] += w * gn2[q];
                    v[14] += h.G * dL_dalpha;
                    v[11] += dL_dz;
                    if (h.use3d) {
                        const float gG = -dL_dG * h.G;
                        const float dpx = (gG * h.sx + dL_dz * Twx) * h.pz_inv, dpy = (gG * h.sy + dL_dz * Twy) * h.pz_inv;
                        const float dpz = -(dpx * h.sx + dpy * h.sy);
                        v[0] += dpx; v[1] += dpy; v[2] += dpz;
                        v[3] = fmaf(xq, dpx, v[3]); v[4] = fmaf(xq, dpy, v[4]); v[5] = fmaf(xq, dpz, v[5]);
                        v[6] = fmaf(yq, dpx, v[6]); v[7] = fmaf(yq, dpy, v[7]); v[8] = fmaf(yq, dpz, v[8]);
                        v[9] = fmaf(dL_dz, h.sx, v[9]); v[10] = fmaf(dL_dz, h.sy, v[10]);
                    } else {
                        const float gG = -dL_dG * h.G * kFilterInvSquare;
                        v[12] = fmaf(gG, h.dx, v[12]);
                        v[13] = fmaf(gG, h.dy, v[13]);
                    }
                }
                dpp_fold_rows<21>(v);   // v[0..5]: value k + vbase, summed over the lanes {l, l^4, l^8, l^12} of the row
                float* orow = &s_out[j][vbase];
#pragma unroll
                for (int k = 0; k < 6; ++k) { v[k] += dpp_mov<0x4E>(v[k]); }   // quad_perm:[2,3,0,1] = lane ^ 2
#pragma unroll
                for (int k = 0; k < 6; ++k) { v[k] += dpp_mov<0xB1>(v[k]); }   // quad_perm:[1,0,3,2] = lane ^ 1: every lane of the quad holds the six totals
                const int i = lane & 3;
                const float t0 = i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
                const float t4 = i == 0 ? v[4] : v[5];
                if (act && vbase + i < 21) lds_add_f32(orow + i, t0);
                if (act && i < 2 && vbase + 4 + i < 21) lds_add_f32(orow + 4 + i, t4);
            }
        }
        pend = ((wrote >> lane) & 1ull) != 0ull; pslot = slot;
    }
    if (pend) flush_record<kGQ, true>(&s_out[lane][0], inst_grads, written, pslot, 0.f, 0.f, false);
}

#define SR_FOR_TILE_SHAPE(F)                                                    \
    if (f.tile_w == 16 && f.tile_h == 16) { F(2, 2); }                          \
    else if (f.tile_w == 8 && f.tile_h == 8) { F(1, 1); }                       \
    else if (f.tile_w == 16 && f.tile_h == 8) { F(2, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 8) { F(4, 1); }                      \
    else if (f.tile_w == 32 && f.tile_h == 16) { F(4, 2); }                     \
    else return hipErrorInvalidValue;

// coop_mode: 0 = by tile count (the cooperative kernel below kCoopBelowTiles tiles of 16x16 with three colour channels), 1 = never, 2 = always (A/B, tests),
// 3 = the row-mapped kernel: the forward must have written CELL-granular hit masks (launch_render_forward flags bit 6)
constexpr int kCoopBelowTiles = 2600;
static inline bool coop_few_tiles(const FrameDev& f, int coop_mode) {
    if (!(f.tile_w == 16 && f.tile_h == 16 && f.colors == 3) || coop_mode == 1) return false;
    return coop_mode == 2 || f.tiles_x * f.tiles_y < kCoopBelowTiles;
}

// flags: bit 0 = quadrant culling on (SR_FLAG_NO_QUADRANT_CULL clear), bit 1 = counter variant (counters != NULL), bit 2 = row-mapped kernel
hipError_t launch_render_backward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                  const float* extra, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                  const float* dL_dallmap, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, bool precomp_color_grads, hipStream_t s,
                                  int coop_mode) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    if (coop_mode == 3) {
        if (!(f.tile_w == 16 && f.tile_h == 16 && f.colors == 3)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(render_backward_rows_kernel, dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, final_T, n_contrib,
                           dL_dcolor, dL_dallmap, hit_mask, inst_grads, written);
        return hipGetLastError();
    }
    if (coop_few_tiles(f, coop_mode)) {   // few tiles (the reference's `-r 4` frames): four quadrant waves per tile instead of one wave
        hipLaunchKernelGGL((render_backward_coop_kernel<3>), dim3(n_tiles), dim3(4 * kWave), 0, s, f, ranges, tile_order, point_list, recs, final_T, n_contrib,
                           dL_dcolor, dL_dallmap, hit_mask, inst_grads, written);
        return hipGetLastError();
    }
    if (!precomp_color_grads && f.tile_w == 16 && f.tile_h == 16 && f.colors != 3) {   // (the reference tile only: elsewhere the sums are formed and nobody reads them)
#define SR_LAUNCH_BWD_NOXG(NCH) hipLaunchKernelGGL((render_backward_kernel<NCH, 2, 2, 1, false>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, extra, \
                                                   final_T, n_contrib, dL_dcolor, dL_dallmap, hit_mask, inst_grads, written)
        if (f.colors == 9) SR_LAUNCH_BWD_NOXG(9); else SR_LAUNCH_BWD_NOXG(6);
#undef SR_LAUNCH_BWD_NOXG
        return hipGetLastError();
    }
#define SR_LAUNCH_BWD(NCH, QX, QY)                                                                                                          \
    hipLaunchKernelGGL((render_backward_kernel<NCH, QX, QY>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, extra, final_T, \
                       n_contrib, dL_dcolor, dL_dallmap, hit_mask, inst_grads, written)
    if (f.tile_w == 16 && f.tile_h == 16) {
#if SR_K7_NINE_BANDED   // A/B (tools/notes_round5_measured.md): the 9-channel K7 as two banded walks of two pixels per lane (158 VGPRs, three waves per SIMD)
        if (f.colors == 9) hipLaunchKernelGGL((render_backward_kernel<9, 2, 1, 2>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, extra,
                                              final_T, n_contrib, dL_dcolor, dL_dallmap, hit_mask, inst_grads, written); else
#endif
        if (f.colors == 9) SR_LAUNCH_BWD(9, 2, 2); else if (f.colors == 6) SR_LAUNCH_BWD(6, 2, 2); else SR_LAUNCH_BWD(3, 2, 2);
    } else if (f.colors != 3 && f.tile_w == 32 && f.tile_h == 16) {   // eight pixels per lane: two banded walks of four (BANDS = 2)
#define SR_LAUNCH_BWD_BANDED(NCH) hipLaunchKernelGGL((render_backward_kernel<NCH, 4, 1, 2>), dim3(n_tiles), dim3(kWave), 0, s, f, ranges, tile_order, point_list, recs, extra, \
                                                     final_T, n_contrib, dL_dcolor, dL_dallmap, hit_mask, inst_grads, written)
        if (f.colors == 9) SR_LAUNCH_BWD_BANDED(9); else SR_LAUNCH_BWD_BANDED(6);
#undef SR_LAUNCH_BWD_BANDED
    } else if (f.colors != 3) {   // 6 / 9 channels on the shapes with up to four pixels per lane (render.hip launch_render_forward)
        if (f.tile_h != 8) return hipErrorInvalidValue;
#define SR_BWD_NC(QX) { if (f.colors == 9) SR_LAUNCH_BWD(9, QX, 1); else SR_LAUNCH_BWD(6, QX, 1); }
        if (f.tile_w == 8) SR_BWD_NC(1) else if (f.tile_w == 16) SR_BWD_NC(2) else if (f.tile_w == 32) SR_BWD_NC(4) else return hipErrorInvalidValue;
#undef SR_BWD_NC
    } else {
#define SR_BWD_SHAPE(QX, QY) SR_LAUNCH_BWD(3, QX, QY)
        SR_FOR_TILE_SHAPE(SR_BWD_SHAPE)
#undef SR_BWD_SHAPE
    }
#undef SR_LAUNCH_BWD
    return hipGetLastError();
}

}  // namespace sr
