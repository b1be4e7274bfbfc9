// common.h -- shared device/host definitions for the surfel rasterizer kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/surfel_raster.h"
#include "../../include/surfel_switches.h"   // the named Appendix-A switches, shared with the oracle

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "these kernels are written for gfx950 (MI355X): wave64, v_permlane16/32_swap, DPP row controls, and the LDS-atomic lane order the sort kernels rank by (checked at run time, api.hip rank_mode)"
#endif

namespace sr {

constexpr int kTile = SR_TILE;
constexpr int kBlock = kTile * kTile;  // 256 threads = 4 wave64
constexpr float kNear = 0.2f;
constexpr float kFar = 100.0f;
constexpr float kFilterSize = 0.707106f;
constexpr float kFilterInvSquare = 2.0f;
constexpr float kCutoff = 3.0f;
constexpr float kAlphaCap = 0.99f;
constexpr float kAlphaFloor = 1.0f / 255.0f;
constexpr float kTStop = 0.0001f;
constexpr uint32_t kCulledKey = 0xFFFFFFFFu;

// Packed per-Gaussian record, 5 x float4 = 80 B, 16-B aligned (one gather = five dwordx4 loads):
//   q0 = Tu.x Tu.y Tu.z Tv.x | q1 = Tv.y Tv.z Tw.x Tw.y | q2 = Tw.z xy.x xy.y opacity
//   q3 = n.x n.y n.z r       | q4 = g b depth radius  (depth = view-space depth; depth and radius are not read by the forward blend:
//   everything K6 stages sits in the first 18 floats -- four dwordx4 + one dwordx2 per entry)
constexpr int kRecQuads = 5;
constexpr int kRecFloats = SR_SPLAT_FLOATS;

// Gradient record written by the blend backward (one per (tile, Gaussian) duplicate, then summed per Gaussian),
// 6 x float4 = 96 B.  The transMat gradient is kept in "moment" form: with dp = dL/dp of the ray-splat cross
// product p = k x l at a pixel (x, y) (pixel coordinates relative to the Gaussian's own centre (cx, cy): round 3),
//   S0 = sum dp, Sx = sum x dp, Sy = sum y dp, Z = sum dL/ddepth * (s.x, s.y, 1)
// are linear in the pixels AND in the tiles, so they can be summed first and turned into dL/dT once per
// Gaussian (K8):  dTu = Tv' x S0 - Tw x Sy,  dTv = S0 x Tu' - Sx x Tw,  dTw = Tu' x Sy - Tv' x Sx + Z - cx dTu - cy dTv
// with Tu' = Tu - cx Tw, Tv' = Tv - cy Tw.
//   slots  0..2 S0 | 3..5 Sx | 6..8 Sy | 9..11 Z | 12,13 d/dxy | 14 d/dopacity | 15..17 d/dnormal | 18..20 d/drgb |
//          21..23 d/d(colour channels 3..5) in the 6-channel variant, else unused.
// Next to the records: one `written` byte per slot, zeroed per call, set by K7 where it stored a record.
constexpr int kGradQuads = 6;
constexpr int kGradFloats = SR_GRAD_FLOATS;

struct FrameDev {
    int W, H, tiles_x, tiles_y;
    int bw_W, bw_H;               // image size as the per-Gaussian backward sees it (SR_BACKWARD_WH_FROM_FOCAL; = W, H by default)
    int tile_w, tile_h;           // pixels per tile: powers of two, multiples of 8 (8x8 quadrant = 1 pixel per lane)
    float inv_tile_w, inv_tile_h; // exact reciprocals
    int sh_degree, sh_coeffs;
    int colors;   // colour channels blended per pixel: 3, or 6 (precomputed colours only)
    int activations;   // SR_ACT_* bits: inputs are the raw (pre-activation) parameters
    float scale_modifier;
    const float* bg;
    const float* view;
    const float* proj;
    const float* campos;
    // emission index of every Gaussian's first duplicate = first[gid] + first_base[gid / kScanTile] (geom buffer, K2); backward only
    const uint32_t* first;
    const uint32_t* first_base;
    float* sh_jac;   // [P][9] d rgb / d centre through the SH view direction (geom buffer; K1 writes it, K8 reads it instead of the SH rows)
    // SR_FLAG_BINNING_CAPACITY (the sync-free forward): the word the capacity guard sets when the frame's duplicates did not fit the
    // caller's binning buffer (nothing was binned or blended then); the per-Gaussian backward kernels treat every Gaussian as invisible --
    // zero gradients, no read beyond the buffers.  NULL in the default (read-back) mode.
    const uint32_t* overflow;
};
constexpr int kScanTile = 2048;   // Gaussians per block of the emission-offset scan
__device__ __forceinline__ uint32_t first_index(const FrameDev& f, uint32_t gid) { return f.first[gid] + f.first_base[gid / kScanTile]; }

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Mask of the lanes where p holds.  (HIP's __ballot takes an int: a bool costs a v_cndmask + v_cmp on the way to the same s_and.)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Inclusive prefix sum over the 64 lanes of a wave: DPP row shifts inside the rows of 16 lanes, then the two row broadcasts
// (six VALU instructions; __shfl_up goes through the LDS crossbar six times).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)v;
}

// ---------------------------------------------------------------------------------------------
// Stable position of a lane's item inside its (wave, digit) run, and the advance of that run's counter: the rank phase of the radix
// sort and of the expanding partition (64 items of one wave at a time, the rows of a wave in order).
//   kAtomic = true : ONE LDS atomic per item.  ds_add_rtn_u32 hands its old values to the lanes of an instruction that hit the same
//                    address in ascending lane order on gfx950 -- not documented, so api.hip runs rank_selfcheck_kernel once per
//                    device before the first sort and only then selects this path;
//   kAtomic = false: the wave64 match-any idiom (`bits` ballots build, for every lane, the mask of the lanes holding the same digit;
//                    rank = popcount(mask & lanes below), one leader per digit advances the counter): documented semantics only,
//                    ~6 VALU instructions per key bit and row.  The fallback, and SR_FLAG_BALLOT_RANKING.
// All 64 lanes must call it; `run` = the calling wave's counters.
// ---------------------------------------------------------------------------------------------
template <bool kAtomic>
__device__ __forceinline__ uint32_t take_run_slot(uint32_t* run, uint32_t d, bool live, int bits) {
    uint32_t pos = 0;
    if (kAtomic) {
        if (live) pos = atomicAdd(&run[d], 1u);
        return pos;
    }
    unsigned long long same = ballot64(live);
    for (int b = 0; b < bits; ++b) {
        const unsigned long long vote = ballot64(((d >> b) & 1u) != 0u);
        same &= ((d >> b) & 1u) ? vote : ~vote;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    if (live) pos = run[d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (live && rank == 0) run[d] += (uint32_t)__popcll(same);
    __builtin_amdgcn_wave_barrier();
    return pos;
}
enum RankMode { kRankUnknown = 0, kRankAtomic = 1, kRankBallot = 2, kRankNone = 3 };

}  // namespace sr
