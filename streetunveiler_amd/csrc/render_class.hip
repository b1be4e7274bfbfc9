// render_class.hip -- the per-class distortion pass (second half): the reference's training iteration renders
// the same view once per semantic class with `render(..., semantic_filter_bit = 1 << k, reverse_semantic = True)` and uses only
// `rend_dist` of each [REF /root/reference/train.py:94-103] -- five full rasterizations (boolean-indexed inputs, K1..K8 each) for
// five distortion maps.  Here K1..K5 run ONCE on all Gaussians; then every tile list is stably partitioned by class
// (class_partition_kernel: the list of tile t becomes [class 0 by depth | class 1 by depth | ...], with a (begin, end) pair per
// (tile, class)), and both blend kernels run one wave per (tile, class) on that class's SUB-LIST alone -- a Gaussian of another class
// is to a class chain what it is to the reference's subset render: absent.  Per class the arithmetic is exactly the class-filtered
// render's: same list order, same alpha / transmittance thresholds, same early termination; contributor numbers count positions
// in the class's own list, as in the subset render.  Every (tile, Gaussian) duplicate gets at most one gradient record, so K8 is
// unchanged.  The class id of a Gaussian arrives in the first colour slot (colors_precomp[:, 0]; there is no colour here).
// (Round 4 walked the whole tile list in every (tile, class) wave of the backward and kept all class chains in one forward wave:
// 67 M staged entries for 8.6 M useful ones, 9.3 GiB fetched per backward launch, a third of the forward's issue slots scalar --
// profiles/r05_train_step_*)
#include "blend_common.h"

namespace sr {

hipError_t launch_zero_bytes(void* p, size_t n, hipStream_t s);   // binning.hip

constexpr int kClassMax = 8;
constexpr uint8_t kNoClass = 255;

// class id bytes: [P] u8 from colors_precomp[:, 0] (negative / >= n_classes / NaN: in no class)
__global__ __launch_bounds__(256) void class_ids_kernel(int P, int n_classes, const float* __restrict__ cols, uint8_t* __restrict__ ids) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float c = cols[3 * (size_t)i];
    const int ci = (int)c;
    ids[i] = (c >= 0.f && ci < n_classes) ? (uint8_t)ci : kNoClass;
}

// ... or from an int32 array (the shared-plan entry points: the colour slot carries real colours there)
__global__ __launch_bounds__(256) void class_ids_i32_kernel(int P, int n_classes, const int32_t* __restrict__ classes, uint8_t* __restrict__ ids) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int ci = classes[i];
    ids[i] = (ci >= 0 && ci < n_classes) ? (uint8_t)ci : kNoClass;
}

// One wave per tile: stable partition of the tile's list by class.  Two sweeps over the list (count, scatter) with wave ballots, four
// chunks of 64 entries in flight per step (the id -> class gather is a dependent load: one chunk at a time the kernel is a chain of
// memory latencies, 0.15 ms at C3); the class bytes of the first kClsCache entries wait in LDS for the second sweep.  Entries of no class
// are dropped.  cls_list shares the tile's span of positions: class c of tile t occupies [cls_ranges[t * n + c].x, .y) inside
// [ranges[t].x, ranges[t].y).
constexpr int kClsCache = 8192, kClsUnroll = 4;
__global__ __launch_bounds__(kWave) void class_partition_kernel(int n_classes, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                                 const uint8_t* __restrict__ ids, uint32_t* __restrict__ cls_list, uint2* __restrict__ cls_ranges) {
    __shared__ uint8_t s_cls[kClsCache];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    uint32_t cnt[kClassMax];
#pragma unroll
    for (int k = 0; k < kClassMax; ++k) cnt[k] = 0;
    for (uint32_t base = 0; base < n; base += kWave * kClsUnroll) {
        // (clamped addresses instead of predicated loads: a predicate makes every load a branch of its own with a wait inside, and the
        // four dependent pairs run one after the other)
        uint32_t c[kClsUnroll], g4[kClsUnroll];
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) g4[u] = point_list[range.x + min(base + u * kWave + lane, n - 1u)];
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) c[u] = ids[g4[u]];
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) c[u] = (base + u * kWave + lane < n) ? c[u] : (uint32_t)kNoClass;
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) {
            const uint32_t j = base + u * kWave + lane;
            if (j < (uint32_t)kClsCache) s_cls[j] = (uint8_t)c[u];   // (kNoClass beyond the end of the list: never read as an entry)
#pragma unroll
            for (int k = 0; k < kClassMax; ++k)
                if (k < n_classes) cnt[k] += (uint32_t)__popcll(ballot64(c[u] == (uint32_t)k));
        }
    }
    uint32_t begin[kClassMax], at = range.x;
#pragma unroll
    for (int k = 0; k < kClassMax; ++k) {
        begin[k] = at;
        if (k < n_classes) {
            if (lane == 0) cls_ranges[(size_t)tile * n_classes + k] = make_uint2(at, at + cnt[k]);
            at += cnt[k];
        }
    }
    __builtin_amdgcn_wave_barrier();   // (one wave: its own LDS writes are visible to it in program order)
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n; base += kWave * kClsUnroll) {
        uint32_t gid[kClsUnroll], c[kClsUnroll];
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) gid[u] = point_list[range.x + min(base + u * kWave + lane, n - 1u)];
        if (base + kWave * kClsUnroll <= (uint32_t)kClsCache) {   // wave-uniform: the whole step is in the LDS cache
#pragma unroll
            for (int u = 0; u < kClsUnroll; ++u) c[u] = (uint32_t)s_cls[min(base + u * kWave + lane, n - 1u)];
        } else {
#pragma unroll
            for (int u = 0; u < kClsUnroll; ++u) c[u] = (uint32_t)ids[gid[u]];
        }
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) c[u] = (base + u * kWave + lane < n) ? c[u] : (uint32_t)kNoClass;
#pragma unroll
        for (int u = 0; u < kClsUnroll; ++u) {
#pragma unroll
            for (int k = 0; k < kClassMax; ++k) {
                if (k >= n_classes) continue;
                const unsigned long long b = ballot64(c[u] == (uint32_t)k);
                if (c[u] == (uint32_t)k) cls_list[begin[k] + (uint32_t)__popcll(b & below)] = gid[u];
                begin[k] += (uint32_t)__popcll(b);
            }
        }
    }
}

// the three record quads the class pass stages (transform rows, centre, opacity); the others stay zero
__device__ __forceinline__ void load_record_geometry(const float4* __restrict__ recs, uint32_t gid, float4 (&q)[kRecQuads]) {
    const float4* r = recs + (size_t)gid * kRecQuads;
    q[0] = r[0]; q[1] = r[1]; q[2] = r[2];
}

// One wave per (tile band, class): QX x 1 quadrants per wave; SPLIT = 2: the reference's 16x16 tile as two 16x8 band waves (two pixels per
// lane), SPLIT = 1: the 8x8 / 16x8 / 32x8 tiles of BASELINE config 5's sweep.  A single transmittance / distortion chain.
// QY = 2 (SPLIT = 1; round 6): the whole 16x16 tile in ONE wave, four pixels per lane -- a class chain keeps five registers per pixel (the colour
// pass: fourteen), so the band split that buys the colour forward its sixth wave per SIMD buys nothing here and costs a second staging of
// every entry (profiles/r05_train_step_hbm_traffic.json: 2.36 x the algorithmic bytes).
template <int QX, int SPLIT, int QY = 1>
__global__ __launch_bounds__(kWave) void class_forward_kernel(FrameDev f, int n_classes, const uint2* __restrict__ cls_ranges, const uint32_t* __restrict__ tile_order,
                                                              const uint32_t* __restrict__ cls_list, const float4* __restrict__ recs,
                                                              float* __restrict__ out_dist,       // [n_classes, H, W]
                                                              float* __restrict__ cls_state,      // [n_classes, 3, H, W]: T_final, M1, M2
                                                              uint32_t* __restrict__ cls_last,    // [n_classes, H, W]: last contributor (position in the class's list, 1-based)
                                                              uint32_t* __restrict__ tile_total,  // [tiles, n_classes]: deepest contributor of the class in the tile (zeroed by the caller)
                                                              uint16_t* __restrict__ hit_mask, int cull) {
    constexpr int NQ = QX * QY;
    static_assert(QY == 1 || SPLIT == 1, "a whole-tile wave is not split into bands");
    constexpr uint32_t kQuadMask = (1u << NQ) - 1u;
    __shared__ float4 s_e[entry_quads<0>()][kWave];
    const int lane = threadIdx.x;
    // the waves of one tile -- its classes, and the SPLIT bands of each (they walk the same sub-list) -- on one XCD
    const int xcd = blockIdx.x % kXcds;
    int k = blockIdx.x / kXcds;
    const int part = k % SPLIT; k /= SPLIT;
    const int cls = k % n_classes;
    int tile = (k / n_classes) * kXcds + xcd;
    if (tile >= f.tiles_x * f.tiles_y) return;
    tile = (int)tile_order[tile];
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8 * SPLIT) + part * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)((tile / f.tiles_x) * (QY * 8 * SPLIT) + QY * SPLIT * 4);
    const float yshift = (float)(part * (QY * 8) - QY * (SPLIT - 1) * 4);
    const int lx = lane & 7, ly = lane >> 3;
    const uint2 range = cls_ranges[(size_t)tile * n_classes + cls];
    const uint32_t n_total = range.y - range.x;
    const float yl0 = (float)(ly - QY * 4) + yshift;   // quadrant row 0; row 1 (QY = 2) adds 8
    float xl[NQ], T[NQ], M1[NQ], M2[NQ], dist[NQ];
    uint32_t lastc[NQ];
    uint32_t done = 0, alive = 0;   // bit q: pixel (lane, q) finished / some pixel of quadrant q still open
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        xl[q] = (float)((q % QX) * 8 + lx - QX * 4);
        const bool outside = !(px < f.W && py < f.H);
        T[q] = 1.f; M1[q] = M2[q] = dist[q] = 0.f; lastc[q] = 0;
        if (outside) done |= 1u << q;
        if (ballot64(!outside) != 0) alive |= 1u << q;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nr[kRecQuads];
#pragma unroll
    for (int i = 0; i < kRecQuads; ++i) nr[i] = zero4;
    // (memory pipeline of the walk as in render.hip's K6: list entries two rounds ahead, records one, hit masks stored behind the next staging)
    uint32_t gid_ahead = 0;
    if ((uint32_t)lane < n_total) load_record_geometry(recs, cls_list[range.x + lane], nr);
    if (kWave + (uint32_t)lane < n_total) gid_ahead = cls_list[range.x + kWave + lane];
    unsigned long long hit_prev[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) hit_prev[q] = 0ull;
    uint32_t n_prev = 0, base_prev = 0;
    auto store_hits = [&]() {
        if ((uint32_t)lane < n_prev) {
            uint32_t hm = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) hm |= (uint32_t)((hit_prev[q] >> lane) & 1ull) << q;
            if (SPLIT == 2) reinterpret_cast<uint8_t*>(hit_mask)[2 * (size_t)(range.x + base_prev + lane) + part] = (uint8_t)hm;
            else if (QY == 2) hit_mask[range.x + base_prev + lane] = (uint16_t)((hm & ((1u << QX) - 1u)) | ((hm >> QX) << 8));   // (decode_hits: a byte per quadrant row)
            else hit_mask[range.x + base_prev + lane] = (uint16_t)hm;
        }
    };
    for (uint32_t base = 0; base < n_total && alive; base += kWave) {
        const uint32_t n = min((uint32_t)kWave, n_total - base);
        uint32_t m = 0;
        wait_vector_memory();   // (everything in flight is a round old: blend_common.h)
        if ((uint32_t)lane < n) m = stage_entry<QX, QY, 0>(nr, zero4, zero4, Xc, Yc, cull & 1, s_e, lane, yshift) & alive & kQuadMask;
        const uint32_t gid = gid_ahead;
        store_hits();
        if (base + 2 * kWave + lane < n_total) gid_ahead = cls_list[range.x + base + 2 * kWave + lane];
        __builtin_amdgcn_sched_barrier(0);
        if (base + kWave + lane < n_total) load_record_geometry(recs, gid, nr);
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long bits = ballot64(m != 0);
        unsigned long long hit[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) hit[q] = 0ull;
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
                const bool valid = intersect(xl[q], yl0 + (float)((q / QX) * 8), e0, e1, e2, e3, h) & !(done & (1u << q));
                if (ballot64(valid) == 0) continue;
                hit[q] |= 1ull << j;
                if (valid) {
                    const float test_T = T[q] * (1.f - h.alpha);
                    if (test_T < kTStop) {
                        done |= 1u << q;  // this entry is NOT blended
                    } else {
                        const float w = h.alpha * T[q];
                        const float A = 1.f - T[q];
                        const float mm = kFN * (1.f - kNear * fast_rcp(h.depth));
                        dist[q] += (mm * mm * A + M2[q] - 2.f * mm * M1[q]) * w;
                        M1[q] += mm * w;
                        M2[q] += mm * mm * w;
                        T[q] = test_T;
                        lastc[q] = contributor;
                    }
                }
                if (ballot64(!(done & (1u << q))) == 0) alive &= ~(1u << q);
            }
        }
        // (entry, quadrant) hit mask for the backward: one byte per band -- stored behind the next round's staging, or behind the walk
#pragma unroll
        for (int q = 0; q < NQ; ++q) hit_prev[q] = hit[q];
        n_prev = n; base_prev = base;
    }
    store_hits();
    // entries behind the point where this band's chain had closed keep whatever hit byte they had: the backward walks to the deepest
    // contributor of EITHER band, but masks every quadrant with its own deepest contributor (`need`), so those bytes are never acted on
    const size_t HW = (size_t)f.H * f.W;
    uint32_t deepest = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = tx0 + (q % QX) * 8 + lx, py = ty0 + (q / QX) * 8 + ly;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            out_dist[cls * HW + pix] = dist[q];
            cls_state[(cls * 3 + 0) * HW + pix] = T[q]; cls_state[(cls * 3 + 1) * HW + pix] = M1[q]; cls_state[(cls * 3 + 2) * HW + pix] = M2[q];
            cls_last[cls * HW + pix] = lastc[q];
        }
        deepest = max(deepest, lastc[q]);
    }
    deepest = wave_max_u32(deepest);
    if (lane == 0 && deepest) atomicMax(&tile_total[(size_t)tile * n_classes + cls], deepest);
}

// Stores the record a lane holds in its row of s_out (the round that just ended) at the duplicate's emission index.  kShared: the records
// already hold what K7 of a colour pass over the SAME binning left there: ADD the fifteen geometry sums (floats 0..14 = quads 0..3) to a
// written record (rec_quads float4 per record: 6, or 7 after a 9-channel K7) and start a new one -- zero-filled -- elsewhere.
template <bool kShared>
__device__ __forceinline__ void class_flush(const float* __restrict__ row, float4* __restrict__ inst_grads, uint8_t* __restrict__ written, uint32_t slot,
                                            float ox, float oy, int rec_quads) {
    constexpr int kGQ = kGradQuads;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* accl = reinterpret_cast<const float4*>(row);
    float4 a0 = accl[0], a1 = accl[1], a2 = accl[2], a3 = accl[3], a4 = accl[4], a5 = accl[5];
    a0.w = fmaf(ox, a0.x, a0.w); a1.x = fmaf(ox, a0.y, a1.x); a1.y = fmaf(ox, a0.z, a1.y);
    a1.z = fmaf(oy, a0.x, a1.z); a1.w = fmaf(oy, a0.y, a1.w); a2.x = fmaf(oy, a0.z, a2.x);
    if (kShared) {
        float4* o = inst_grads + (size_t)slot * rec_quads;
        if (written[slot]) {
            const float4 p0 = o[0], p1 = o[1], p2 = o[2], p3 = o[3];
            o[0] = make_float4(p0.x + a0.x, p0.y + a0.y, p0.z + a0.z, p0.w + a0.w); o[1] = make_float4(p1.x + a1.x, p1.y + a1.y, p1.z + a1.z, p1.w + a1.w);
            o[2] = make_float4(p2.x + a2.x, p2.y + a2.y, p2.z + a2.z, p2.w + a2.w); o[3] = make_float4(p3.x + a3.x, p3.y + a3.y, p3.z + a3.z, p3.w + a3.w);
        } else {
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;
            for (int k = kGQ; k < rec_quads; ++k) o[k] = zero4;
            written[slot] = 1;
        }
    } else {
        float4* o = inst_grads + (size_t)slot * kGQ;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;
        written[slot] = 1;
    }
}

// One wave per (tile, class): the blend backward of the class-filtered render with the distortion gradient as the only upstream
// gradient (no colour, depth, normal, alpha, median terms): psi = dLw, Z = sum_{k>i} w_k dLw_k  (render_bwd.hip, K7).  Walks the
// class's sub-list back to front from the deepest contributor any pixel of the tile has.
// kShared: the records already hold what K7 of a colour pass over the SAME binning left there (sr_class_backward_shared): this pass ADDS its
// 15 sums to a written record (rec_quads float4 per record: 6, or 7 after a 9-channel K7) and starts a new one -- zero-filled -- elsewhere.
template <int QX, int QY, bool kShared>
#ifndef SR_CLASS_BWD_WAVES
#define SR_CLASS_BWD_WAVES 4
#endif
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(QX * QY <= 4 ? SR_CLASS_BWD_WAVES : 2, QX * QY <= 4 ? SR_CLASS_BWD_WAVES : 2)))
void class_backward_kernel(FrameDev f, int n_classes, const uint2* __restrict__ cls_ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ cls_list,
                           const float4* __restrict__ recs, const float* __restrict__ cls_state, const uint32_t* __restrict__ cls_last,
                           const uint32_t* __restrict__ tile_total, const float* __restrict__ dL_ddist, const uint16_t* __restrict__ hit_mask,
                           float4* __restrict__ inst_grads, uint8_t* __restrict__ written, int rec_quads) {
    constexpr int NQ = QX * QY, kGQ = kGradQuads;
    __shared__ float4 s_e[entry_quads<0>()][kWave];
    __shared__ __attribute__((aligned(16))) float s_out[kWave][kGQ * 4];
    const int lane = threadIdx.x;
    // the classes of one tile on one XCD (their pixels' state and the records of their Gaussians' neighbours share that L2)
    const int xcd = blockIdx.x % kXcds, kk = blockIdx.x / kXcds;
    const int cls = kk % n_classes;
    int tile = (kk / n_classes) * kXcds + xcd;
    if (tile >= f.tiles_x * f.tiles_y) return;
    tile = (int)tile_order[tile];
    const uint32_t total = tile_total[(size_t)tile * n_classes + cls];   // deepest position of the class's list any pixel of the tile needs
    if (total == 0) return;
    const int tx0 = (tile % f.tiles_x) * (QX * 8), ty0 = (tile / f.tiles_x) * (QY * 8);
    const float Xc = (float)(tx0 + QX * 4), Yc = (float)(ty0 + QY * 4);
    const int lx = lane & 7, ly = lane >> 3;
    const uint32_t first_pos = cls_ranges[(size_t)tile * n_classes + cls].x;
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
#pragma unroll
    for (int i = 0; i < kRecQuads; ++i) nr[i] = zero4;
    // Memory pipeline of the walk as in render_bwd.hip's K7 (round 6): the list entry two rounds ahead, the record one round ahead and RAW
    // (first[] / first_base[] as two registers, the hit mask undecoded), the records of a round stored behind the staging of the next one.
    uint32_t nfirst = 0, nfbase = 0, nhraw = 0, gid_ahead = 0;
#define SR_CLASS_FETCH(GID, POS) { const uint32_t gid_ = (GID); load_record_geometry(recs, gid_, nr);                                   \
        nr[4].w = reinterpret_cast<const float*>(recs + (size_t)gid_ * kRecQuads + 4)[3];   /* the radius (emission_index) */ \
        nfirst = f.first[gid_]; nfbase = f.first_base[gid_ / kScanTile]; nhraw = hit_mask[(POS)]; }
    if ((uint32_t)((rounds - 1) * kWave + lane) < total) SR_CLASS_FETCH(cls_list[first_pos + (rounds - 1) * kWave + lane], first_pos + (rounds - 1) * kWave + lane)
    if (rounds > 1) gid_ahead = cls_list[first_pos + (rounds - 2) * kWave + lane];
    bool pend = false;
    uint32_t pslot = 0;
    float pox = 0.f, poy = 0.f;
    for (int rd = rounds - 1; rd >= 0; --rd) {
        const uint32_t rbase = (uint32_t)rd * kWave;
        const uint32_t n = min((uint32_t)kWave, total - rbase);
        uint32_t m = 0, slot = 0;
        float ox = 0.f, oy = 0.f;
        wait_vector_memory();
        if ((uint32_t)lane < n) {
            (void)stage_entry<QX, QY, 0>(nr, zero4, zero4, Xc, Yc, 0, s_e, lane);
            slot = emission_index(nr, nfirst + nfbase, tile % f.tiles_x, tile / f.tiles_x, f);
            uint32_t need = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) need |= (rbase + lane < quad_last[q]) ? (1u << q) : 0u;
            m = decode_hits<QX, QY>((uint16_t)nhraw) & need;
            // (clamped into the image: the moments of a splat whose centre projects far off-screen are taken about the nearest image point)
            const float mx = nr[2].y - Xc, my = nr[2].z - Yc;   // the staged centre relative to the tile centre: same bits as stage_entry's
            ox = -fminf(fmaxf(mx, -Xc), (float)(f.W - 1) - Xc); oy = -fminf(fmaxf(my, -Yc), (float)(f.H - 1) - Yc);
        }
        if (rd > 0) {
            SR_CLASS_FETCH(gid_ahead, first_pos + rbase - kWave + lane)
            if (rd > 1) gid_ahead = cls_list[first_pos + rbase - 2 * kWave + lane];
        }
        if (pend) class_flush<kShared>(&s_out[lane][0], inst_grads, written, pslot, pox, poy, rec_quads);
        {
            float4* z = reinterpret_cast<float4*>(&s_out[lane][0]);
#pragma unroll
            for (int k = 0; k < kGQ; ++k) z[k] = zero4;
        }
        unsigned long long bits = ballot64(m != 0);
        const unsigned long long wrote = bits;   // every entry with a forward hit gets a record (see K7)
        while (bits) {
            const int j = 63 - __clzll((long long)bits);
            bits &= ~(1ull << j);
            const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);
            const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j];
            const uint32_t cidx = rbase + (uint32_t)j;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = 0.f;
                if (k < 15) asm volatile("" : "+v"(v[k]));   // opaque zero for the live accumulators (see K7); v[15] stays constant zero
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
                const float tot = wave_reduce16(v);   // record floats 15..23 (colour, normal) stay the zeros s_out was reset to
                if (reduce16_holds_total(lane)) s_out[j][reduce16_index(lane)] = tot;
            }
        }
        pend = ((wrote >> lane) & 1ull) != 0ull; pslot = slot; pox = ox; poy = oy;
    }
    if (pend) class_flush<kShared>(&s_out[lane][0], inst_grads, written, pslot, pox, poy, rec_quads);
#undef SR_CLASS_FETCH
}

hipError_t launch_class_partition(int P, int n_tiles, int n_classes, const float* class_cols, const int32_t* class_i32, const uint2* ranges,
                                  const uint32_t* point_list, uint8_t* ids, uint32_t* cls_list, uint2* cls_ranges, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    if (n_classes < 1 || n_classes > kClassMax) return hipErrorInvalidValue;
    if (P > 0 && class_i32) hipLaunchKernelGGL(class_ids_i32_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, n_classes, class_i32, ids);
    else if (P > 0) hipLaunchKernelGGL(class_ids_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, n_classes, class_cols, ids);
    // (P == 0: every list is empty, the kernel only writes the empty (begin, end) pairs)
    hipLaunchKernelGGL(class_partition_kernel, dim3(n_tiles), dim3(kWave), 0, s, n_classes, ranges, point_list, ids, cls_list, cls_ranges);
    return hipGetLastError();
}

hipError_t launch_class_forward(const FrameDev& f, int n_classes, const uint2* cls_ranges, const uint32_t* tile_order, const uint32_t* cls_list,
                                const float4* recs, float* out_dist, float* cls_state, uint32_t* cls_last, uint32_t* tile_total, uint16_t* hit_mask,
                                int cull, hipStream_t s) {
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    if (n_classes < 1 || n_classes > kClassMax) return hipErrorInvalidValue;
    const bool two_bands = f.tile_h == 16 && (f.tile_w == 16 || f.tile_w == 32);   // 16x16 and 32x16: two band waves per (tile, class)
    if (!two_bands && !(f.tile_h == 8 && (f.tile_w == 8 || f.tile_w == 16 || f.tile_w == 32))) return hipErrorInvalidValue;
    hipError_t e = launch_zero_bytes(tile_total, sizeof(uint32_t) * (size_t)n_tiles * n_classes, s);   // (binning.hip: not hipMemsetAsync)
    if (e != hipSuccess) return e;
#ifndef SR_CLASS_FWD_WHOLE_TILE
#define SR_CLASS_FWD_WHOLE_TILE 1   // the reference's 16x16 tile as ONE wave per (tile, class) instead of two band waves (A/B: 0)
#endif
    const bool whole = SR_CLASS_FWD_WHOLE_TILE && two_bands && f.tile_w == 16;
    const int split = (two_bands && !whole) ? 2 : 1;
    const dim3 grid((unsigned)((n_tiles + kXcds - 1) / kXcds * kXcds) * (unsigned)(split * n_classes));
#define SR_CF_SHAPE(QX, SPLIT) hipLaunchKernelGGL((class_forward_kernel<QX, SPLIT>), grid, dim3(kWave), 0, s, f, n_classes, cls_ranges, tile_order, cls_list, recs, out_dist, cls_state, cls_last, tile_total, hit_mask, cull)
    if (whole) hipLaunchKernelGGL((class_forward_kernel<2, 1, 2>), grid, dim3(kWave), 0, s, f, n_classes, cls_ranges, tile_order, cls_list, recs, out_dist, cls_state, cls_last, tile_total, hit_mask, cull);
    else if (two_bands) { if (f.tile_w == 16) SR_CF_SHAPE(2, 2); else SR_CF_SHAPE(4, 2); }
    else if (f.tile_w == 8) SR_CF_SHAPE(1, 1); else if (f.tile_w == 16) SR_CF_SHAPE(2, 1); else SR_CF_SHAPE(4, 1);
#undef SR_CF_SHAPE
    return hipGetLastError();
}

hipError_t launch_class_backward(const FrameDev& f, int n_classes, const uint2* cls_ranges, const uint32_t* tile_order, const uint32_t* cls_list,
                                 const float4* recs, const float* cls_state, const uint32_t* cls_last, const uint32_t* tile_total,
                                 const float* dL_ddist, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, int shared_rec_quads, hipStream_t s) {
    // shared_rec_quads = 0: this pass owns the records (6 quads each); 6 / 7: it adds to those of a colour pass (kShared)
    const int n_tiles = f.tiles_x * f.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    if (n_classes < 1 || n_classes > kClassMax) return hipErrorInvalidValue;
    const bool two_rows = f.tile_h == 16 && (f.tile_w == 16 || f.tile_w == 32);
    if (!two_rows && !(f.tile_h == 8 && (f.tile_w == 8 || f.tile_w == 16 || f.tile_w == 32))) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_tiles + kXcds - 1) / kXcds * kXcds) * (unsigned)n_classes);
#define SR_CB_SHAPE(QX, QY) { if (shared_rec_quads) hipLaunchKernelGGL((class_backward_kernel<QX, QY, true>), grid, dim3(kWave), 0, s, f, n_classes, cls_ranges, tile_order, cls_list, recs, cls_state, cls_last, tile_total, dL_ddist, hit_mask, inst_grads, written, shared_rec_quads); \
                             else hipLaunchKernelGGL((class_backward_kernel<QX, QY, false>), grid, dim3(kWave), 0, s, f, n_classes, cls_ranges, tile_order, cls_list, recs, cls_state, cls_last, tile_total, dL_ddist, hit_mask, inst_grads, written, (int)kGradQuads); }
    if (two_rows) { if (f.tile_w == 16) SR_CB_SHAPE(2, 2) else SR_CB_SHAPE(4, 2) }   // (32x16: eight pixels per lane, two waves per SIMD)
    else if (f.tile_w == 8) SR_CB_SHAPE(1, 1) else if (f.tile_w == 16) SR_CB_SHAPE(2, 1) else SR_CB_SHAPE(4, 1)
#undef SR_CB_SHAPE
    return hipGetLastError();
}

}  // namespace sr
