// blend_common.h -- device helpers shared by the blend kernels (render.hip: K6 / decision dump, render_bwd.hip: K7; render_class.hip: the
// per-class distortion pass): staging of list entries into the tile-local form, exact quadrant culling, the ray-splat test,
// the emission index of a duplicate, and the wave-level transpose-reduction of the per-entry gradient sums.
#pragma once
#include "common.h"

namespace sr {

constexpr int kWave = 64;
constexpr int kXcds = 8;   // MI355X: 8 accelerator dies, workgroup i runs on XCD i % 8
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kFN = kFar / (kFar - kNear);

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima), placed by hand at the
// top of a round of the list walks: everything in flight there was issued a round earlier, and a wait that sits on EVERY control-flow
// path lets the compiler's own counter tracking start the round from zero -- a wait inside the (lane < n) staging branch leaves the
// skip path "pending", and the next use of a prefetched register then waits again, behind whatever store was issued in between.
__device__ __forceinline__ void wait_vector_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// counters of the K6 counter variant (SrFrame.blend_counters, caller-owned, 16 x u64): [0] entries staged, [1] entries with a
// non-zero quadrant mask, [2] quadrant tests executed, [3] quadrant tests with >= 1 valid lane, [4] valid (pixel, entry) pairs,
// [5] / [6] tests with a valid pixel in rows 0-3 / rows 4-7 of the quadrant, [7] entries with a valid pixel anywhere in the tile,
// [8] (entry, 4x4 cell) pairs with a valid pixel, [9] sum over (round of 64 entries, quadrant) of the busiest cell's pair count,
// [10] / [11] the same two with the pairs an octagon-vs-cell culling at staging would KEEP (hits and misses) instead of the exact hits,
// [12] / [13] ... an octagon + oriented-box culling would keep, [14] exact (entry, cell) pairs that the oriented box would DROP (must be 0)

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
// octagon of the entry's footprint in tile-local coordinates: lo / hi of x, y, x + y, x - y.  Returns 0 = empty footprint, 2 = unbounded
// (the cutoff disc reaches the camera plane, or something is degenerate: keep the entry everywhere), 1 = bounds valid.
__device__ __forceinline__ int octagon_bounds(const float Tu[3], const float Tv[3], const float Tw[3], float mx, float my, float opacity,
                                              float (&lo)[4], float (&hi)[4]) {
    float thr = 2.f * __logf(255.f * opacity);
    thr = thr * 1.01f + 0.01f;
    if (thr <= 0.f) return 0;
    const float c22 = thr * (Tw[0] * Tw[0] + Tw[1] * Tw[1]) - Tw[2] * Tw[2];
    if (!(c22 < 0.f)) return 2;  // the cutoff disc reaches the camera plane: unbounded footprint
    const float c00 = thr * (Tu[0] * Tu[0] + Tu[1] * Tu[1]) - Tu[2] * Tu[2];
    const float c01 = thr * (Tu[0] * Tv[0] + Tu[1] * Tv[1]) - Tu[2] * Tv[2];
    const float c11 = thr * (Tv[0] * Tv[0] + Tv[1] * Tv[1]) - Tv[2] * Tv[2];
    const float c02 = thr * (Tu[0] * Tw[0] + Tu[1] * Tw[1]) - Tu[2] * Tw[2];
    const float c12 = thr * (Tv[0] * Tw[0] + Tv[1] * Tw[1]) - Tv[2] * Tw[2];
    const float inv = fast_rcp(c22);
    const float r = __builtin_amdgcn_sqrtf(0.5f * thr);
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
    return 1;
}
// does the octagon reach the pixel-centre rectangle [x0, x1] x [y0, y1] (0.3 px of slack for float rounding)?
__device__ __forceinline__ bool octagon_reaches(const float (&lo)[4], const float (&hi)[4], float x0, float x1, float y0, float y1) {
    const float m = 0.3f;
    x0 -= m; y0 -= m; x1 += m; y1 += m;
    return !(lo[0] > x1 || hi[0] < x0 || lo[1] > y1 || hi[1] < y0 || lo[2] > x1 + y1 || hi[2] < x0 + y0 || lo[3] > x1 - y0 || hi[3] < x0 - y1);
}

template <int QX, int QY>
__device__ __forceinline__ uint32_t quadrant_mask(const float Tu[3], const float Tv[3], const float Tw[3], float mx, float my,
                                                  float opacity, float yshift) {
    constexpr uint32_t kAll = (1u << (QX * QY)) - 1u;
    float lo[4], hi[4];
    const int kind = octagon_bounds(Tu, Tv, Tw, mx, my, opacity, lo, hi);
    if (kind != 1) return kind ? kAll : 0u;
    // the wave's quadrants sit `yshift` below the local origin: shift the bounds instead of the (compile-time) rectangles
    lo[1] -= yshift; hi[1] -= yshift; lo[2] -= yshift; hi[2] -= yshift; lo[3] += yshift; hi[3] += yshift;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < QX * QY; ++q) {   // quadrant (q % QX, q / QX) of the tile, coordinates relative to the tile centre
        const float x0 = (float)((q % QX) * 8 - QX * 4), y0 = (float)((q / QX) * 8 - QY * 4);
        if (octagon_reaches(lo, hi, x0, x0 + 7.f, y0, y0 + 7.f)) mask |= 1u << q;
    }
    return mask;
}
// The same test against the four 4x4 cells of each of the wave's quadrants: bit 4 q + c = cell c of quadrant q, cell c at (c & 1, c >> 1)
// inside its quadrant (the row-mapped forward blend: one 16-lane row per cell).  (Two more slabs along the principal axes of the footprint
// ellipse -- cell_mask16's oriented box -- keep 6 % fewer (entry, cell) pairs and cost as much at staging as they save: measured, not kept.)
template <int QX, int QY>
__device__ __forceinline__ uint32_t cell_mask_rows(const float Tu[3], const float Tv[3], const float Tw[3], float mx, float my,
                                                   float opacity, float yshift) {
    constexpr uint32_t kAll = (QX * QY >= 8) ? 0xFFFFFFFFu : (1u << (4 * QX * QY)) - 1u;
    float lo[4], hi[4];
    const int kind = octagon_bounds(Tu, Tv, Tw, mx, my, opacity, lo, hi);
    if (kind != 1) return kind ? kAll : 0u;
    lo[1] -= yshift; hi[1] -= yshift; lo[2] -= yshift; hi[2] -= yshift; lo[3] += yshift; hi[3] += yshift;
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < QX * QY; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = (float)((q % QX) * 8 + (c & 1) * 4 - QX * 4), y0 = (float)((q / QX) * 8 + (c >> 1) * 4 - QY * 4);
            if (!octagon_reaches(lo, hi, x0, x0 + 3.f, y0, y0 + 3.f)) continue;
            mask |= 1u << (4 * q + c);
        }
    return mask;
}
// (counter variant only) the same test against the sixteen 4x4 cells of a 16x16 tile: bit 4 q + c = cell c of quadrant q, cell c at
// (c & 1, c >> 1) inside its quadrant.  Bits 16..31: the same with two more slabs, along the principal axes of the footprint ellipse
// (an oriented box around ellipse and filter disc): what a tighter -- still conservative -- cell culling would keep.
__device__ __forceinline__ uint32_t cell_mask16(const float Tu[3], const float Tv[3], const float Tw[3], float mx, float my, float opacity) {
    float lo[4], hi[4];
    const int kind = octagon_bounds(Tu, Tv, Tw, mx, my, opacity, lo, hi);
    if (kind != 1) return kind ? 0xFFFFFFFFu : 0u;
    // oriented box: centre c, axes u / v, half extents E1 / E2 (support of the ellipse {rho3d <= thr} = sqrt(d^T S d), S from the dual conic)
    float thr = 2.f * __logf(255.f * opacity);
    thr = thr * 1.01f + 0.01f;
    const float c22 = thr * (Tw[0] * Tw[0] + Tw[1] * Tw[1]) - Tw[2] * Tw[2];
    const float c00 = thr * (Tu[0] * Tu[0] + Tu[1] * Tu[1]) - Tu[2] * Tu[2];
    const float c01 = thr * (Tu[0] * Tv[0] + Tu[1] * Tv[1]) - Tu[2] * Tv[2];
    const float c11 = thr * (Tv[0] * Tv[0] + Tv[1] * Tv[1]) - Tv[2] * Tv[2];
    const float c02 = thr * (Tu[0] * Tw[0] + Tu[1] * Tw[1]) - Tu[2] * Tw[2];
    const float c12 = thr * (Tv[0] * Tw[0] + Tv[1] * Tw[1]) - Tv[2] * Tw[2];
    const float inv = fast_rcp(c22), inv2 = inv * inv;
    const float cx = c02 * inv, cy = c12 * inv;
    const float sxx = (c02 * c02 - c22 * c00) * inv2, syy = (c12 * c12 - c22 * c11) * inv2, sxy = (c02 * c12 - c22 * c01) * inv2;
    const float a = 0.5f * (sxx - syy), r = __builtin_amdgcn_sqrtf(a * a + sxy * sxy), mid = 0.5f * (sxx + syy);
    float ux = a >= 0.f ? a + r : sxy, uy = a >= 0.f ? sxy : r - a;
    const float ul = ux * ux + uy * uy;
    const float un = ul > 0.f ? __builtin_amdgcn_rsqf(ul) : 0.f;
    ux = ul > 0.f ? ux * un : 1.f; uy = ul > 0.f ? uy * un : 0.f;
    const float rd = __builtin_amdgcn_sqrtf(0.5f * thr);
    const float ox = mx - cx, oy = my - cy;
    const float E1 = fmaxf(__builtin_amdgcn_sqrtf(fmaxf(mid + r, 0.f)) * 1.01f, fabsf(ox * ux + oy * uy) + rd);
    const float E2 = fmaxf(__builtin_amdgcn_sqrtf(fmaxf(mid - r, 0.f)) * 1.01f, fabsf(oy * ux - ox * uy) + rd);
    const bool box_ok = E1 == E1 && E2 == E2;   // anything degenerate: the box says nothing
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = (float)((q % 2) * 8 + (c & 1) * 4 - 8), y0 = (float)((q / 2) * 8 + (c >> 1) * 4 - 8);
            if (!octagon_reaches(lo, hi, x0, x0 + 3.f, y0, y0 + 3.f)) continue;
            mask |= 1u << (4 * q + c);
            const float rx = x0 + 1.5f - cx, ry = y0 + 1.5f - cy, h = 1.8f;
            const float R = h * (fabsf(ux) + fabsf(uy));
            const bool out = fabsf(rx * ux + ry * uy) > E1 + R || fabsf(ry * ux - rx * uy) > E2 + R;
            if (!(box_ok && out)) mask |= 1u << (16 + 4 * q + c);
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
template <int NC> constexpr int entry_quads() { return NC == 9 ? 7 : (NC == 0 ? 4 : 6); }   // NC = 0: geometry only (e0..e3: the per-class distortion pass)

__device__ __forceinline__ float4 load_extra(const float* __restrict__ colors6, uint32_t gid, int first) {
    const float* c = colors6 + 6 * (size_t)gid + first;
    return make_float4(c[0], c[1], c[2], 0.f);
}

template <int QX, int QY, int NC>
__device__ __forceinline__ uint32_t stage_entry(const float4 (&q)[kRecQuads], const float4 ex, const float4 ey, float Xc, float Yc, int cull,
                                                float4 (*s_e)[kWave], int slot, float yshift = 0.f, uint32_t* cells16 = nullptr, uint32_t* cells_rows = nullptr) {
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
    // p.z = x A.z + y B.z + C.z vanishes at EVERY pixel when a scale (or the quaternion) is exactly zero -- the three coefficients are then
    // exact zeros, here as in the reference's cross(k, l), whose `if (p.z == 0) continue` drops such a splat altogether: it is staged
    // with opacity 0 and never passes the alpha test.  (An isolated p.z == 0 of a healthy splat is rounding noise: see intersect().)
    const bool plane_degenerate = A[2] == 0.f && B[2] == 0.f && C[2] == 0.f;
    s_e[3][slot] = make_float4(mx, my, plane_degenerate ? 0.f : opacity, ex.z);
    if (NC != 0) {
        s_e[4][slot] = make_float4(q[3].x, q[3].y, q[3].z, q[3].w);   // n.xyz, c0  (component-wise: a whole-quad copy of an array element keeps the array in scratch)
        s_e[5][slot] = make_float4(q[4].x, q[4].y, ex.x, ex.y);   // c1, c2 | c3, c4
    }
    if (NC == 9) s_e[6][slot] = make_float4(ey.x, ey.y, ey.z, 0.f);
    if (cells16) *cells16 = cell_mask16(Tu, Tv, Tw, mx, my, opacity);   // (counter variant: what a 4x4-cell culling would keep)
    if (cells_rows) {   // (row-mapped forward: per-cell bits; a quadrant is visited if one of its cells is)
        *cells_rows = cell_mask_rows<QX, QY>(Tu, Tv, Tw, mx, my, opacity, yshift);
        uint32_t qm = 0;
#pragma unroll
        for (int q = 0; q < QX * QY; ++q) qm |= ((*cells_rows >> (4 * q)) & 15u) ? (1u << q) : 0u;
        return qm;
    }
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
    const bool use2d = !(rho3d <= rho2d);   // (ONE compare serves the depth select here and the path branch of the backward)
    h.use3d = !use2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = use2d ? e2.w : fmaf(h.sx, e2.y, fmaf(h.sy, e2.z, e2.w));
    h.G = __builtin_amdgcn_exp2f(rho * (-0.5f * kLog2e));   // exp(-rho / 2)
    h.alpha = fminf(kAlphaCap, e3.z * h.G);
    // (the reference also skips on `power > 0`: never true -- rho3d and rho2d are sums of squares, fminf drops a NaN operand, and a NaN
    // power fails that test as well -- so the comparison is not evaluated here)
    // The reference's `if (p.z == 0) continue` (Appendix A.4; SR_REFERENCE_PZ_SKIP = 1, the shipped value) is evaluated on the staged cross
    // product p.z = x A.z + y B.z + C.z.  For a healthy splat p.z == 0 means the pixel's ray is parallel to the splat's plane: a set of
    // measure zero in exact arithmetic, and in float32 a coin toss among the pairs whose p.z is pure rounding noise -- which of them land on
    // exactly 0 depends on the operation order, so this expression and the reference's cross(k, l).z can skip DIFFERENT pairs of that
    // noise set (the oracle's margin walk classifies them; profiles/r05_parity_c3.json counts them).  A splat whose p.z vanishes
    // identically (a zero scale) has three exact-zero coefficients and is skipped everywhere, here as there.
    // SR_REFERENCE_PZ_SKIP = 0 (variant pz_zero_through_filter) is the exact-arithmetic rule instead: with ppz = 0, rcp gives inf, rho3d is
    // inf or NaN, the comparison above takes the screen-space path and fminf drops rho3d -- the pair is blended through its 2-D filter
    // footprint, what exact arithmetic does with the astronomically large rho3d of a nearly parallel ray (identically-zero splats are
    // dropped at staging: stage_entry).  sx, sy, pz_inv are only read on the ray-splat path.
#if SR_REFERENCE_PZ_SKIP
    return !(h.depth < kNear) & !(h.alpha < kAlphaFloor) & !(ppz == 0.f);   // upstream's per-pair `if (p.z == 0) continue`, on THIS cross product
#else
    return !(h.depth < kNear) & !(h.alpha < kAlphaFloor);
#endif
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

// the 18 floats the forward blend stages (everything but depth and radius): four dwordx4 + one dwordx2
__device__ __forceinline__ void load_record18(const float4* __restrict__ recs, uint32_t gid, float4 (&q)[kRecQuads]) {
    const float4* r = recs + (size_t)gid * kRecQuads;
#pragma unroll
    for (int k = 0; k < kRecQuads - 1; ++k) q[k] = r[k];
    const float2 t = *reinterpret_cast<const float2*>(r + (kRecQuads - 1));
    q[kRecQuads - 1].x = t.x; q[kRecQuads - 1].y = t.y;
}
__device__ __forceinline__ void load_record(const float4* __restrict__ recs, uint32_t gid, float4 (&q)[kRecQuads]) {
    const float4* r = recs + (size_t)gid * kRecQuads;
#pragma unroll
    for (int k = 0; k < kRecQuads; ++k) q[k] = r[k];
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
// Returns the 64-lane totals of the 24 values, one per lane: lane l ends up with the total of value index
//   reduce24_index(l) = (bit1(l) ? 2 : bit4(l)) + 3 bit5(l) + 6 bit2(l) + 12 bit3(l),
// valid in the lanes with bit0 clear and not (bit1 and bit4) -- 24 lanes.
// DPP first: the two in-row levels (lane ^ 8, lane ^ 4) are transposing folds done with bank-masked DPP adds -- the rotated operand
// rides in the add itself (1.8 ns per wave instruction against 3.9 + 1.05 for a lane swap plus its add), two instructions per pair
// of registers and no selects: the first writes the banks that keep `a`, the second the banks that keep `b`.  24 -> 12 -> 6 registers
// cost 36 DPP adds; only then do the cross-row levels run on 6 registers (5 swaps instead of 18), and the quad levels on 2.
// Measured against the swap-first order it replaces (18 swaps + 9 DPP moves + 27 adds + 14 selects): tools/ubench/reduce_ubench.hip.
// One asm block: the DPP reads of a register follow its last write by >= 7 instructions inside the block (the hardware wants 2
// wait states between a VALU write and a DPP read, and the compiler's hazard recogniser does not look into inline asm); the
// s_nop in front covers whatever wrote the inputs.
// NV = 24, or 21: values 21..23 are not in use (three colour channels) -- they need no reset and no fold; their slots of the result hold
// copies of other totals, which nothing reads.
// Preconditions: ALL 64 lanes active (DPP without bound_ctrl leaves the destination of a lane whose source is disabled untouched: the
// callers run the fold outside any divergent region -- K7's entry loop is wave-uniform); the twelve outputs are early-clobber ("+&v"):
// they are overwritten while the input-only registers are still to be read, so no input may share a register with an output.
template <int NV>
__device__ __forceinline__ void dpp_fold_rows(float (&v)[24]) {
    static_assert(NV == 24 || NV == 21, "24 values, or 21 with the last three unused");
    if (NV == 24) {
        asm volatile(
        "s_nop 1\n"
        // level lane^8, lanes 0-7 of every row: v[k] += v[k] of lane^8 (all lanes written; lanes 8-15 are overwritten next)
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        // ... lanes 8-15 (banks 2, 3) take value k + 12 instead
        "v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %8, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %9, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %10, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %11, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        // level lane^4: lanes with bit 2 clear (banks 0, 2) keep register k (partner = lane + 4: row_ror:12), lanes with bit 2 set
        // (banks 1, 3) take register k + 6 (partner = lane - 4: row_ror:4)
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %4, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %0, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %1, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %2, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %4, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %5, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "s_nop 1\n"
        : "+&v"(v[0]), "+&v"(v[1]), "+&v"(v[2]), "+&v"(v[3]), "+&v"(v[4]), "+&v"(v[5]), "+&v"(v[6]), "+&v"(v[7]), "+&v"(v[8]), "+&v"(v[9]), "+&v"(v[10]), "+&v"(v[11])
        : "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]), "v"(v[21]), "v"(v[22]), "v"(v[23]));
    } else {
        asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %8, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %4, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %5, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %0, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %1, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %2, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %3, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %4, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %5, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "s_nop 1\n"
        : "+&v"(v[0]), "+&v"(v[1]), "+&v"(v[2]), "+&v"(v[3]), "+&v"(v[4]), "+&v"(v[5]), "+&v"(v[6]), "+&v"(v[7]), "+&v"(v[8]), "+&v"(v[9]), "+&v"(v[10]), "+&v"(v[11])
        : "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]));
    }
}
// ---------------------------------------------------------------------------------------------
// Zeros from the LDS (round 6).  The backward kernels clear 15..24 accumulator registers per list entry; as v_mov instructions that was 8 % of K7's
// vector instructions.  The same registers filled by broadcast ds_read_b128 of a small block of zeros cost the vector unit nothing: the loads issue
// down the LDS port -- which these kernels leave three-quarters idle -- beside the entry's own four, and the compiler's counter tracking lets the
// ray-splat test start while they are in flight.  Four copies of the block, picked by the entry slot: an address that changes per entry keeps the
// loads inside the entry loop (a loop-invariant address would be hoisted and the zeros copied with -- v_mov).  The block is written through an
// opaque register, so its contents are not a constant the loads could be folded into.  C3: K7 1.670 -> 1.615 ms, bit-identical (same-box A/B).
// ---------------------------------------------------------------------------------------------
constexpr int kZeroCopies = 4;
template <int kQuads>
__device__ __forceinline__ void lds_zeros_init(float4 (*s_zero)[kZeroCopies], int lane) {   // by every lane of (one of) the workgroup's waves; same-wave readers need no barrier
    float zo = 0.f;
    asm volatile("" : "+v"(zo));
    if (lane < kQuads * kZeroCopies) s_zero[lane / kZeroCopies][lane % kZeroCopies] = make_float4(zo, zo, zo, zo);
}
template <int kN, int kLen>
__device__ __forceinline__ void lds_zeros_load(const float4 (*s_zero)[kZeroCopies], int slot, float (&v)[kLen]) {   // v[0 .. kN) from the LDS, the rest plain zeros
    const int zi = slot & (kZeroCopies - 1);
#pragma unroll
    for (int k = 0; k < (kN + 3) / 4; ++k) {
        const float4 z = s_zero[k][zi];
        if (4 * k < kN) v[4 * k] = z.x;
        if (4 * k + 1 < kN) v[4 * k + 1] = z.y;
        if (4 * k + 2 < kN) v[4 * k + 2] = z.z;
        if (4 * k + 3 < kN) v[4 * k + 3] = z.w;
    }
#pragma unroll
    for (int k = kN; k < kLen; ++k) v[k] = 0.f;
}

// LDS float add without a return value (ds_add_f32): one wave's adds to an address land in program order
__device__ __forceinline__ void lds_add_f32(float* p, float x) { (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int reduce24_index(int l) { return ((l & 2) ? 2 : ((l >> 4) & 1)) + 3 * ((l >> 5) & 1) + 6 * ((l >> 2) & 1) + 12 * ((l >> 3) & 1); }
__device__ __forceinline__ bool reduce24_holds_total(int l) { return (l & 1) == 0 && !((l & 2) && (l & 16)); }
template <int NV = 24>
__device__ __forceinline__ float wave_reduce24(float (&v)[24], int lane) {
    dpp_fold_rows<NV>(v);                          // v[0..5]: value k + 6 bit2 + 12 bit3, summed over the lanes {l, l^4, l^8, l^12}
#pragma unroll
    for (int k = 0; k < 3; ++k) fold32(v[k], v[k + 3]);    // + 3 bit5, summed over both halves
    float z = 0.f;
    fold16(v[0], v[1]);                        // bit4 clear: value 0 (+ ...), bit4 set: value 1; summed over all four rows
    fold16(v[2], z);                           // bit4 clear: value 2; bit4 set: nothing
    float a = v[0], b = v[2];
    // the two quad levels as four DPP adds (written out: left to the compiler the last level became v_mov 0 + v_mov_dpp + v_add per value, the add
    // sunk into the caller's `holds_total` branch); the s_nops are the two wait states between a vector write and its DPP read
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"   // lane ^ 2
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"   // lane ^ 1: every lane of a quad holds the totals
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        : "+v"(a), "+v"(b));
    return (lane & 2) ? b : a;
}

// The same reduction for 16 values (the per-class distortion backward: 15 live accumulators): 16 -> 8 -> 4 registers in-row (24 bank-masked
// DPP adds), two half folds and one row-pair fold on 4 registers (3 swaps), two quad levels on one.  26 DPP + 3 swaps against the 40 + 5
// of wave_reduce24.  Afterwards EVERY lane holds the 64-lane total of value reduce16_index(lane); the lanes with bits 0 and 1 clear
// (16 of them) cover all sixteen.  Same preconditions as dpp_fold_rows (all 64 lanes active; >= 7 instructions between a write and
// its DPP read inside the block: 8 here).
__device__ __forceinline__ int reduce16_index(int l) { return ((l >> 4) & 1) + 2 * ((l >> 5) & 1) + 4 * ((l >> 2) & 1) + 8 * ((l >> 3) & 1); }
__device__ __forceinline__ bool reduce16_holds_total(int l) { return (l & 3) == 0; }
__device__ __forceinline__ float wave_reduce16(float (&v)[16]) {
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %0, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %1, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %2, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %3, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n"
        "s_nop 1\n"
        : "+&v"(v[0]), "+&v"(v[1]), "+&v"(v[2]), "+&v"(v[3]), "+&v"(v[4]), "+&v"(v[5]), "+&v"(v[6]), "+&v"(v[7])
        : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    fold32(v[0], v[2]); fold32(v[1], v[3]);   // lanes < 32: values 0 / 1 (+ 4 bit2 + 8 bit3), lanes >= 32: values 2 / 3
    fold16(v[0], v[1]);                       // bit4 clear: value 0 or 2, bit4 set: value 1 or 3; summed over all four rows
    float a = v[0];
    asm volatile(   // (written out as in wave_reduce24)
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"   // lane ^ 2
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"   // lane ^ 1
        : "+v"(a));
    return a;
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


}  // namespace sr
