// preprocess.hip -- K1 (per-Gaussian forward), K8 (per-Gaussian backward), K9 (mark_visible).
//
// Compiled with -ffp-contract=off: every value that feeds an integer output (radius, tile
// rectangle, depth-key bits) is computed with the same single mul/add sequence as the CPU oracle,
// so radii / tiles_touched / sort keys are bit-exact against it.  These kernels are HBM-streaming
// (one thread per Gaussian, ~230 B in / ~90 B out), the extra non-fused ops are free.
//
// Behavioural contract: SURVEY.md Appendix A.2 / A.6; operator inputs
// [REF /root/reference/gaussian_renderer/__init__.py:56-138]; SH polynomial
// [REF /root/reference/utils/sh_utils.py:57-112]; quaternion convention
// [REF /root/reference/utils/general_utils.py:85-98].
#include "common.h"

namespace sr {

__device__ __constant__ float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                            -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                            0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                            -0.5900435899266435f};

__device__ __forceinline__ void quat_to_R(const float4 q, float R[9]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Parameter activations of the reference's GaussianModel, fused into K1 / K8 on request
// [REF scene/gaussian_model.py:63-75: scaling exp, opacity sigmoid, rotation torch.nn.functional.normalize].
__device__ __forceinline__ float2 activate_scales(float2 s, int act) {
    if (act & SR_ACT_EXP_SCALES) { s.x = expf(s.x); s.y = expf(s.y); }
    return s;
}
__device__ __forceinline__ float2 load_scales(const float* __restrict__ scales, int i, int act) {
    return activate_scales(reinterpret_cast<const float2*>(scales)[i], act);
}
__device__ __forceinline__ float4 activate_rotation(float4 q, int act, float& norm) {
    norm = 1.f;
    if (act & SR_ACT_NORMALIZE_ROTATIONS) {
        norm = fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
        q.x /= norm; q.y /= norm; q.z /= norm; q.w /= norm;
    }
    return q;
}
__device__ __forceinline__ float4 load_rotation(const float* __restrict__ rotations, int i, int act, float& norm) {
    return activate_rotation(reinterpret_cast<const float4*>(rotations)[i], act, norm);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// B = viewport^T * full_projection (3x4): rows map a world point to (px*w, py*w, w).
__device__ __forceinline__ void build_B(const float* __restrict__ proj, int W, int H, float B[12]) {
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float cw = 0.5f * (float)(W - 1), ch = 0.5f * (float)(H - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a0 = proj[4 * k + 0], a1 = proj[4 * k + 1], a3 = proj[4 * k + 3];
        B[0 + k] = hw * a0 + cw * a3;
        B[4 + k] = hh * a1 + ch * a3;
        B[8 + k] = a3;
    }
}

__device__ __forceinline__ float dot3(const float* a, float x, float y, float z) { return (a[0] * x + a[1] * y) + a[2] * z; }

// SH rows are 3*M floats per Gaussian (192 B for M = 16).  One thread per Gaussian reading its own row
// would touch 64 different cache lines per load instruction; instead a block of kPreBlock threads moves its rows
// through LDS with fully coalesced 16-B accesses (48 KB in, and for K8 48 KB out), and each thread works on
// its row in LDS.  Row stride 52 floats (48 + 4 pad) keeps per-thread ds_read_b128 conflict-free.
constexpr int kShRowFloats = 48;   // fast path: M == 16
constexpr int kShLdsStride = 52;
#ifndef SR_PRE_BLOCK
#define SR_PRE_BLOCK 128
#endif
constexpr int kPreBlock = SR_PRE_BLOCK;   // threads (= Gaussians) per block of K1 / K8 / the SH expansion

__device__ __forceinline__ void sh_rows_to_lds(const float* __restrict__ shs, int base, int P, float* s_sh, int tid) {
    const int nrows = min(kPreBlock, P - base);
    const int nvec = nrows * (kShRowFloats / 4);
    const float4* src = reinterpret_cast<const float4*>(shs + (size_t)base * kShRowFloats);
    float4* dst = reinterpret_cast<float4*>(s_sh);
    for (int f = tid; f < nvec; f += kPreBlock) {
        const int row = f / 12, c4 = f - row * 12;
        dst[row * (kShLdsStride / 4) + c4] = src[f];
    }
}

__device__ __forceinline__ void sh_rows_from_lds(float* __restrict__ out, int base, int P, const float* s_sh, int tid) {
    const int nrows = min(kPreBlock, P - base);
    const int nvec = nrows * (kShRowFloats / 4);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)base * kShRowFloats);
    const float4* src = reinterpret_cast<const float4*>(s_sh);
    for (int f = tid; f < nvec; f += kPreBlock) {
        const int row = f / 12, c4 = f - row * 12;
        dst[f] = src[row * (kShLdsStride / 4) + c4];
    }
}

// colour = SH(deg, sh, dir) + 0.5, clamped at 0 (flags recorded)  [REF utils/sh_utils.py:57-112]
template <class Row>
__device__ __forceinline__ void sh_to_rgb(int deg, const Row sh, float dx, float dy, float dz, float rgb[3], uint8_t& clamped) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float res = kSH_C0 * sh[c];
        if (deg > 0) {
            const float x = dx, y = dy, z = dz;
            res = res - kSH_C1 * y * sh[3 + c] + kSH_C1 * z * sh[6 + c] - kSH_C1 * x * sh[9 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + kSH_C2[0] * xy * sh[12 + c] + kSH_C2[1] * yz * sh[15 + c] +
                      kSH_C2[2] * (2.f * zz - xx - yy) * sh[18 + c] + kSH_C2[3] * xz * sh[21 + c] +
                      kSH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    res = res + kSH_C3[0] * y * (3.f * xx - yy) * sh[27 + c] + kSH_C3[1] * xy * z * sh[30 + c] +
                          kSH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + c] +
                          kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
                          kSH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + c] +
                          kSH_C3[5] * z * (xx - yy) * sh[42 + c] + kSH_C3[6] * x * (xx - 3.f * yy) * sh[45 + c];
                }
            }
        }
        res += 0.5f;
        if (res < 0.f) clamped |= (uint8_t)(1u << c);
        rgb[c] = fmaxf(res, 0.f);
    }
}

// ---- K1's SH rows in two halves ---------------------------------------------------------------------------------------------
// K1 is bound by how many waves fit beside its LDS (26 KB per block of 128 rows: three waves per SIMD; with half of that it runs
// 40 % slower).  So the rows go through LDS one half at a time -- coefficients 0..7, then 8..15, 96 B each -- and the colour (and
// the direction Jacobian) are accumulated across the two halves.  The colour keeps the exact operation order of sh_to_rgb(): the
// cut falls between two terms of its left-to-right sum.
constexpr int kShHalfFloats = 24, kShHalfStride = 28;   // 7 quads per row: an odd quad stride keeps per-thread ds_read_b128 conflict-free
// `skip[row]` != 0: the row is not fetched (a Gaussian behind the near plane or masked out: its colour is never evaluated)
__device__ __forceinline__ void sh_half_load(const float* __restrict__ shs, int base, int P, int half, int tid, const uint8_t* skip, float4 (&r)[6]) {
    const int nrows = min(kPreBlock, P - base);
    const float4* src = reinterpret_cast<const float4*>(shs + (size_t)base * kShRowFloats);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int f = tid + k * kPreBlock, row = f / 6, c4 = f - row * 6;
        r[k] = (row < nrows && !skip[row]) ? src[row * 12 + half * 6 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void sh_half_store(float* s_sh, int tid, const float4 (&r)[6]) {
    float4* dst = reinterpret_cast<float4*>(s_sh);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int f = tid + k * kPreBlock, row = f / 6, c4 = f - row * 6;
        dst[row * (kShHalfStride / 4) + c4] = r[k];
    }
}
// first half of sh_to_rgb(): coefficients 0..7 (lo[3 k + c])
__device__ __forceinline__ void sh_to_rgb_lo(int deg, const float* lo, float x, float y, float z, float res[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = kSH_C0 * lo[c];
        if (deg > 0) {
            r = r - kSH_C1 * y * lo[3 + c] + kSH_C1 * z * lo[6 + c] - kSH_C1 * x * lo[9 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + kSH_C2[0] * xy * lo[12 + c] + kSH_C2[1] * yz * lo[15 + c] + kSH_C2[2] * (2.f * zz - xx - yy) * lo[18 + c] +
                    kSH_C2[3] * xz * lo[21 + c];
            }
        }
        res[c] = r;
    }
}
// second half: coefficients 8..15 (hi[3 (k - 8) + c]), then the offset and the clamp
__device__ __forceinline__ void sh_to_rgb_hi(int deg, const float* hi, float x, float y, float z, const float res[3], float rgb[3], uint8_t& clamped) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = res[c];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y;
            r = r + kSH_C2[4] * (xx - yy) * hi[c];
            if (deg > 2) {
                r = r + kSH_C3[0] * y * (3.f * xx - yy) * hi[3 + c] + kSH_C3[1] * xy * z * hi[6 + c] +
                    kSH_C3[2] * y * (4.f * zz - xx - yy) * hi[9 + c] +
                    kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * hi[12 + c] +
                    kSH_C3[4] * x * (4.f * zz - xx - yy) * hi[15 + c] +
                    kSH_C3[5] * z * (xx - yy) * hi[18 + c] + kSH_C3[6] * x * (xx - 3.f * yy) * hi[21 + c];
            }
        }
        r += 0.5f;
        if (r < 0.f) clamped |= (uint8_t)(1u << c);
        rgb[c] = fmaxf(r, 0.f);
    }
}
// the two halves of sh_dir_jacobian(): d[3 c + axis] accumulates d rgb[c] / d dir; the second half projects and scales
__device__ __forceinline__ void sh_dir_jacobian_lo(int deg, const float* lo, float x, float y, float z, float d[9]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (deg > 0) {
            dx = -kSH_C1 * lo[9 + c]; dy = -kSH_C1 * lo[3 + c]; dz = kSH_C1 * lo[6 + c];
            if (deg > 1) {
                dx += (kSH_C2[0] * y) * lo[12 + c] + (-2.f * kSH_C2[2] * x) * lo[18 + c] + (kSH_C2[3] * z) * lo[21 + c];
                dy += (kSH_C2[0] * x) * lo[12 + c] + (kSH_C2[1] * z) * lo[15 + c] + (-2.f * kSH_C2[2] * y) * lo[18 + c];
                dz += (kSH_C2[1] * y) * lo[15 + c] + (4.f * kSH_C2[2] * z) * lo[18 + c] + (kSH_C2[3] * x) * lo[21 + c];
            }
        }
        d[3 * c] = dx; d[3 * c + 1] = dy; d[3 * c + 2] = dz;
    }
}
__device__ __forceinline__ void sh_dir_jacobian_hi(int deg, const float* hi, float x, float y, float z, float len, const float d[9], float J[9]) {
#pragma clang fp contract(fast)
    const float inv_len = 1.f / len;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float dx = d[3 * c], dy = d[3 * c + 1], dz = d[3 * c + 2];
        if (deg > 1) {
            dx += (2.f * kSH_C2[4] * x) * hi[c];
            dy += (-2.f * kSH_C2[4] * y) * hi[c];
            if (deg > 2) {
                dx += (6.f * kSH_C3[0] * xy) * hi[3 + c] + (kSH_C3[1] * yz) * hi[6 + c] + (-2.f * kSH_C3[2] * xy) * hi[9 + c] +
                      (-6.f * kSH_C3[3] * xz) * hi[12 + c] + (kSH_C3[4] * (-3.f * xx + 4.f * zz - yy)) * hi[15 + c] +
                      (2.f * kSH_C3[5] * xz) * hi[18 + c] + (3.f * kSH_C3[6] * (xx - yy)) * hi[21 + c];
                dy += (3.f * kSH_C3[0] * (xx - yy)) * hi[3 + c] + (kSH_C3[1] * xz) * hi[6 + c] + (kSH_C3[2] * (-3.f * yy + 4.f * zz - xx)) * hi[9 + c] +
                      (-6.f * kSH_C3[3] * yz) * hi[12 + c] + (-2.f * kSH_C3[4] * xy) * hi[15 + c] + (-2.f * kSH_C3[5] * yz) * hi[18 + c] +
                      (-6.f * kSH_C3[6] * xy) * hi[21 + c];
                dz += (kSH_C3[1] * xy) * hi[6 + c] + (8.f * kSH_C3[2] * yz) * hi[9 + c] + (3.f * kSH_C3[3] * (2.f * zz - xx - yy)) * hi[12 + c] +
                      (8.f * kSH_C3[4] * xz) * hi[15 + c] + (kSH_C3[5] * (xx - yy)) * hi[18 + c];
            }
        }
        const float nd = (x * dx + y * dy) + z * dz;
        J[3 * c + 0] = (dx - x * nd) * inv_len; J[3 * c + 1] = (dy - y * nd) * inv_len; J[3 * c + 2] = (dz - z * nd) * inv_len;
    }
}

// d rgb[c] / d centre through the SH view direction, J[3 c + k] = ((d_c - dir (dir . d_c)) / len)[k] with d_c = d rgb[c] / d dir
// [REF the direction half of computeColorFromSH's backward, Appendix A.6].  K1 evaluates it next to the colour, from the SH row it
// holds anyway, so that K8 needs 36 B per Gaussian instead of reading the 192-B row again.
template <class Row>
__device__ __forceinline__ void sh_dir_jacobian(int deg, const Row sh, float x, float y, float z, float len, float J[9]) {
#pragma clang fp contract(fast)   // not part of the bit-exact contract with the oracle (gradients are compared with a tolerance)
    const float inv_len = 1.f / len;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (deg > 0) {
            dx = -kSH_C1 * sh[9 + c]; dy = -kSH_C1 * sh[3 + c]; dz = kSH_C1 * sh[6 + c];
            if (deg > 1) {
                dx += (kSH_C2[0] * y) * sh[12 + c] + (-2.f * kSH_C2[2] * x) * sh[18 + c] + (kSH_C2[3] * z) * sh[21 + c] + (2.f * kSH_C2[4] * x) * sh[24 + c];
                dy += (kSH_C2[0] * x) * sh[12 + c] + (kSH_C2[1] * z) * sh[15 + c] + (-2.f * kSH_C2[2] * y) * sh[18 + c] + (-2.f * kSH_C2[4] * y) * sh[24 + c];
                dz += (kSH_C2[1] * y) * sh[15 + c] + (4.f * kSH_C2[2] * z) * sh[18 + c] + (kSH_C2[3] * x) * sh[21 + c];
                if (deg > 2) {
                    dx += (6.f * kSH_C3[0] * xy) * sh[27 + c] + (kSH_C3[1] * yz) * sh[30 + c] + (-2.f * kSH_C3[2] * xy) * sh[33 + c] +
                          (-6.f * kSH_C3[3] * xz) * sh[36 + c] + (kSH_C3[4] * (-3.f * xx + 4.f * zz - yy)) * sh[39 + c] +
                          (2.f * kSH_C3[5] * xz) * sh[42 + c] + (3.f * kSH_C3[6] * (xx - yy)) * sh[45 + c];
                    dy += (3.f * kSH_C3[0] * (xx - yy)) * sh[27 + c] + (kSH_C3[1] * xz) * sh[30 + c] + (kSH_C3[2] * (-3.f * yy + 4.f * zz - xx)) * sh[33 + c] +
                          (-6.f * kSH_C3[3] * yz) * sh[36 + c] + (-2.f * kSH_C3[4] * xy) * sh[39 + c] + (-2.f * kSH_C3[5] * yz) * sh[42 + c] +
                          (-6.f * kSH_C3[6] * xy) * sh[45 + c];
                    dz += (kSH_C3[1] * xy) * sh[30 + c] + (8.f * kSH_C3[2] * yz) * sh[33 + c] + (3.f * kSH_C3[3] * (2.f * zz - xx - yy)) * sh[36 + c] +
                          (8.f * kSH_C3[4] * xz) * sh[39 + c] + (kSH_C3[5] * (xx - yy)) * sh[42 + c];
                }
            }
        }
        const float nd = (x * dx + y * dy) + z * dz;
        J[3 * c + 0] = (dx - x * nd) * inv_len; J[3 * c + 1] = (dy - y * nd) * inv_len; J[3 * c + 2] = (dz - z * nd) * inv_len;
    }
}

// SH adjoint: dsh[k] = basis_k(dir) * dL/drgb.  Coefficients above the active degree get 0.
template <class RowOut>
__device__ __forceinline__ void sh_basis_adjoint(int deg, int M, RowOut dsh, float x, float y, float z, const float gcol[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = gcol[c];
        dsh[c] = kSH_C0 * g;
        if (deg > 0) {
            dsh[3 + c] = -kSH_C1 * y * g; dsh[6 + c] = kSH_C1 * z * g; dsh[9 + c] = -kSH_C1 * x * g;
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dsh[12 + c] = kSH_C2[0] * xy * g; dsh[15 + c] = kSH_C2[1] * yz * g;
                dsh[18 + c] = kSH_C2[2] * (2.f * zz - xx - yy) * g;
                dsh[21 + c] = kSH_C2[3] * xz * g; dsh[24 + c] = kSH_C2[4] * (xx - yy) * g;
                if (deg > 2) {
                    dsh[27 + c] = kSH_C3[0] * y * (3.f * xx - yy) * g;
                    dsh[30 + c] = kSH_C3[1] * xy * z * g;
                    dsh[33 + c] = kSH_C3[2] * y * (4.f * zz - xx - yy) * g;
                    dsh[36 + c] = kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                    dsh[39 + c] = kSH_C3[4] * x * (4.f * zz - xx - yy) * g;
                    dsh[42 + c] = kSH_C3[5] * z * (xx - yy) * g;
                    dsh[45 + c] = kSH_C3[6] * x * (xx - 3.f * yy) * g;
                }
            }
        }
    }
    const int used = (deg + 1) * (deg + 1);
    for (int k = used * 3; k < M * 3; ++k) dsh[k] = 0.f;
}

// The splat -> pixel transform (Appendix A.2 step 3) and the AABB centre (step 5), as K1 evaluates them.  K8 calls the SAME functions on the
// same inputs instead of reading them back from the 80-B record (this translation unit is compiled with -ffp-contract=off, so the same
// expressions give the same bits): 205 MB less to fetch per frame at C3.
__device__ __forceinline__ void splat_transform(const float B[12], float px, float py, float pz, const float R[9], float su, float sv, float Tm[9]) {
    const float L0[3] = {R[0] * su, R[3] * su, R[6] * su};
    const float L1[3] = {R[1] * sv, R[4] * sv, R[7] * sv};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* b = B + 4 * r;
        Tm[3 * r + 0] = dot3(b, L0[0], L0[1], L0[2]);
        Tm[3 * r + 1] = dot3(b, L1[0], L1[1], L1[2]);
        Tm[3 * r + 2] = dot3(b, px, py, pz) + b[3];
    }
}
__device__ __forceinline__ float bbox_cutoff(float act_opacity) {
#if SR_TIGHTBBOX   // upstream config.h TIGHTBBOX: the extent follows the opacity
    return sqrtf(fmaxf(9.f + 2.f * logf(act_opacity), 0.000001f));
#else
    (void)act_opacity;
    return kCutoff;
#endif
}
struct SplatBox { float dist, cx, cy, ex, ey; };
__device__ __forceinline__ SplatBox splat_box(const float Tm[9], float cutoff) {
    const float* Tu = Tm; const float* Tv = Tm + 3; const float* Tw = Tm + 6;
    const float c2 = cutoff * cutoff;
    const float t0 = c2, t1 = c2, t2 = -1.f;
    SplatBox b;
    b.dist = ((Tw[0] * Tw[0]) * t0 + (Tw[1] * Tw[1]) * t1) + (Tw[2] * Tw[2]) * t2;
    const float inv = 1.f / b.dist;
    const float f0 = inv * t0, f1 = inv * t1, f2 = inv * t2;
    b.cx = ((f0 * Tu[0]) * Tw[0] + (f1 * Tu[1]) * Tw[1]) + (f2 * Tu[2]) * Tw[2];
    b.cy = ((f0 * Tv[0]) * Tw[0] + (f1 * Tv[1]) * Tw[1]) + (f2 * Tv[2]) * Tw[2];
    const float tx = ((f0 * Tu[0]) * Tu[0] + (f1 * Tu[1]) * Tu[1]) + (f2 * Tu[2]) * Tu[2];
    const float ty = ((f0 * Tv[0]) * Tv[0] + (f1 * Tv[1]) * Tv[1]) + (f2 * Tv[2]) * Tv[2];
    const float hx = b.cx * b.cx - tx, hy = b.cy * b.cy - ty;
    b.ex = sqrtf(fmaxf(1e-4f, hx)); b.ey = sqrtf(fmaxf(1e-4f, hy));
    return b;
}

// ---------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------
template <bool kLdsSH>
__global__ __launch_bounds__(kPreBlock) void preprocess_forward_kernel(
    int P, FrameDev f, const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const float* __restrict__ transMat_precomp,
    const uint8_t* __restrict__ mask, float4* __restrict__ recs, uint32_t* __restrict__ depth_keys,
    uint32_t* __restrict__ tiles_touched, uint2* __restrict__ rect, uint8_t* __restrict__ clamped, int32_t* __restrict__ radii) {
    // SH half rows (kShHalfStride floats per Gaussian) while the colour is evaluated, then the outgoing records + sh_jac rows (20 + 9)
    constexpr int kK1LdsFloats = kPreBlock * (kShHalfStride > kRecFloats + 9 ? kShHalfStride : kRecFloats + 9);
    __shared__ __attribute__((aligned(16))) float s_sh[kLdsSH ? kK1LdsFloats : 4];
    __shared__ uint8_t s_skip[kLdsSH ? kPreBlock : 4];
    const int tid = threadIdx.x, base = blockIdx.x * kPreBlock;
    const int i = base + tid;
    // every per-Gaussian input is requested up front, in front of the SH staging and its barrier: one memory round trip instead of four
    // dependent ones (SH rows | centre | rotation + scales | opacity) -- this kernel waits on memory at three waves per SIMD
    float px = 0.f, py = 0.f, pz = 0.f, raw_opacity = 0.f;
    float4 raw_rot = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 raw_scales = make_float2(0.f, 0.f);
    bool kept = true;
    if (i < P) {
        px = means3D[3 * i]; py = means3D[3 * i + 1]; pz = means3D[3 * i + 2];
        raw_opacity = opacities[i];
        if (!transMat_precomp) { raw_rot = reinterpret_cast<const float4*>(rotations)[i]; raw_scales = reinterpret_cast<const float2*>(scales)[i]; }
        if (mask) kept = mask[i] != 0;   // masked out == not there (same as boolean-indexing the inputs)
    }
    float4 half_b[6];   // the second halves of the block's SH rows: in flight during the geometry
    if (kLdsSH) {
        // The 192-B SH row is 5/6 of a Gaussian's input bytes, and a camera inside the scene has most Gaussians behind it: rows whose
        // Gaussian fails the near-plane test (or the mask) are not fetched.  The test needs the centre alone, so the rows are requested
        // one short dependent step after everything else.
        const float* vw = f.view;
        s_skip[tid] = !(i < P && kept && ((vw[2] * px + vw[6] * py) + vw[10] * pz) + vw[14] > kNear);
        __syncthreads();
        float4 half_a[6];
        sh_half_load(shs, base, P, 0, tid, s_skip, half_a);
        sh_half_load(shs, base, P, 1, tid, s_skip, half_b);
        sh_half_store(s_sh, tid, half_a);
        __syncthreads();
    }
    const bool in_range = i < P;
    if (!kLdsSH && !in_range) return;   // (with LDS staging every thread stays for the barriers below)
    // defaults for a culled Gaussian
    int32_t out_radius = 0;
    uint32_t out_tiles = 0, out_key = kCulledKey;
    uint2 out_rect = make_uint2(0u, 0u);   // tile rectangle for the duplicate emission: minx | miny << 16, width | height << 16
    uint8_t out_clamped = 0;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0, q4 = q0;

    const float* v = f.view;
    const float vx = ((v[0] * px + v[4] * py) + v[8] * pz) + v[12];
    const float vy = ((v[1] * px + v[5] * py) + v[9] * pz) + v[13];
    const float vz = ((v[2] * px + v[6] * py) + v[10] * pz) + v[14];
    bool alive = in_range && vz > kNear && kept;
    bool need_sh = false;   // the colour comes from the SHs: evaluated behind the geometry, across the two LDS halves
    float sdx = 0.f, sdy = 0.f, sdz = 0.f, slen = 1.f, sradius = 0.f;
    if (alive) {
        float Tm[9], nrm[3];
        if (transMat_precomp) {
#pragma unroll
            for (int k = 0; k < 9; ++k) Tm[k] = transMat_precomp[9 * (size_t)i + k];
            nrm[0] = 0.f; nrm[1] = 0.f; nrm[2] = 1.f;
        } else {
            float B[12], R[9];
            build_B(f.proj, f.W, f.H, B);
            float qn;
            quat_to_R(activate_rotation(raw_rot, f.activations, qn), R);
            const float2 s = activate_scales(raw_scales, f.activations);
            splat_transform(B, px, py, pz, R, f.scale_modifier * s.x, f.scale_modifier * s.y, Tm);
            const float nx = R[2], ny = R[5], nz = R[8];
            nrm[0] = (v[0] * nx + v[4] * ny) + v[8] * nz;
            nrm[1] = (v[1] * nx + v[5] * ny) + v[9] * nz;
            nrm[2] = (v[2] * nx + v[6] * ny) + v[10] * nz;
        }
        const float cosv = -((vx * nrm[0] + vy * nrm[1]) + vz * nrm[2]);
        alive = cosv != 0.f;
        const float mult = cosv > 0.f ? 1.f : -1.f;
        nrm[0] *= mult; nrm[1] *= mult; nrm[2] *= mult;

        const float* Tu = Tm; const float* Tv = Tm + 3; const float* Tw = Tm + 6;
        const float cutoff = bbox_cutoff((f.activations & SR_ACT_SIGMOID_OPACITY) ? sigmoidf(raw_opacity) : raw_opacity);
        const SplatBox box = splat_box(Tm, cutoff);
        alive = alive && (box.dist != 0.f);
        if (alive) {
            const float cx = box.cx, cy = box.cy, ex = box.ex, ey = box.ey;
#if SR_RADIUS_FILTER_FLOOR
            const float radius = ceilf(fmaxf(fmaxf(ex, ey), cutoff * kFilterSize));
#else
            const float radius = ceilf(fmaxf(ex, ey));   // older upstream revisions: no low-pass floor on the radius
#endif
            // tile sizes are powers of two: multiplying by the exact reciprocal == the reference's division
            int minx = (int)((cx - radius) * f.inv_tile_w), miny = (int)((cy - radius) * f.inv_tile_h);
            int maxx = (int)((cx + radius + (float)(f.tile_w - 1)) * f.inv_tile_w);
            int maxy = (int)((cy + radius + (float)(f.tile_h - 1)) * f.inv_tile_h);
            minx = min(f.tiles_x, max(0, minx)); maxx = min(f.tiles_x, max(0, maxx));
            miny = min(f.tiles_y, max(0, miny)); maxy = min(f.tiles_y, max(0, maxy));
            const int area = (maxx - minx) * (maxy - miny);
            if (area > 0) {
                float rgb[3] = {0.f, 0.f, 0.f};
                if (colors_precomp && f.colors != 9) {   // 9 channels: rgb from the SHs, the six extra channels go straight to the blend kernels
                    const float* c = colors_precomp + (size_t)f.colors * i;   // channels 3..5 (if any) are read by the blend kernels
                    rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
                } else {
                    float dx = px - f.campos[0], dy = py - f.campos[1], dz = pz - f.campos[2];
                    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
                    dx /= len; dy /= len; dz /= len;
                    if (kLdsSH) {
                        need_sh = true; sdx = dx; sdy = dy; sdz = dz; slen = len; sradius = radius;   // (q4 is completed below)
                    } else {
                        float J[9];
                        sh_to_rgb(f.sh_degree, shs + (size_t)i * f.sh_coeffs * 3, dx, dy, dz, rgb, out_clamped);
                        if (f.sh_jac) {   // (NULL with SR_FLAG_FORWARD_ONLY: K8 is the only reader)
                            sh_dir_jacobian(f.sh_degree, shs + (size_t)i * f.sh_coeffs * 3, dx, dy, dz, len, J);
#pragma unroll
                            for (int k = 0; k < 9; ++k) f.sh_jac[9 * (size_t)i + k] = J[k];
                        }
                    }
                }
                out_radius = (int32_t)radius;
                out_tiles = (uint32_t)area;
                out_rect = make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)(maxx - minx) | ((uint32_t)(maxy - miny) << 16));
                out_key = __float_as_uint(vz);
                q0 = make_float4(Tu[0], Tu[1], Tu[2], Tv[0]);
                q1 = make_float4(Tv[1], Tv[2], Tw[0], Tw[1]);
                q2 = make_float4(Tw[2], cx, cy, (f.activations & SR_ACT_SIGMOID_OPACITY) ? sigmoidf(raw_opacity) : raw_opacity);
                q3 = make_float4(nrm[0], nrm[1], nrm[2], rgb[0]);
                q4 = make_float4(rgb[1], rgb[2], vz, radius);
            }
        }
    }
    if (kLdsSH) {
        const float* row = s_sh + tid * kShHalfStride;
        float res[3] = {0.f, 0.f, 0.f}, dd[9];
        const bool want_jac = f.sh_jac != nullptr;   // (NULL with SR_FLAG_FORWARD_ONLY: K8 is the only reader of the 36-B rows)
        if (need_sh) {
            sh_to_rgb_lo(f.sh_degree, row, sdx, sdy, sdz, res);
            if (want_jac) sh_dir_jacobian_lo(f.sh_degree, row, sdx, sdy, sdz, dd);
        }
        __syncthreads();   // every thread is done with the first halves
        sh_half_store(s_sh, tid, half_b);
        __syncthreads();
        float J[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (need_sh) {
            float rgb[3];
            sh_to_rgb_hi(f.sh_degree, row, sdx, sdy, sdz, res, rgb, out_clamped);
            if (want_jac) sh_dir_jacobian_hi(f.sh_degree, row, sdx, sdy, sdz, slen, dd, J);
            q3.w = rgb[0];
            q4 = make_float4(rgb[1], rgb[2], q4.z, sradius);
        }
        // The 80-B records and the 36-B sh_jac rows leave through LDS: a block's rows are contiguous in memory, so its 128 lanes store
        // consecutive 16-B chunks instead of five (nine) stores per lane at an 80-B (36-B) stride -- partial sectors that the L2 does not
        // merge for free: K1 0.245 -> 0.219 ms with the records alone.  The staging buffer is free by now.
        __syncthreads();
        float4* s_rec = reinterpret_cast<float4*>(s_sh);
        float* s_jac = s_sh + kPreBlock * kRecFloats;
        // Rows of culled Gaussians are never read (the lists hold visible Gaussians only, K8 skips the others).  A 128-B line of the block's
        // output whose rows are ALL culled is not written: seen from inside a scene most rows are such rows.  Whole lines only -- leaving
        // single 16-B chunks out costs more in partial sectors than it saves (K1 0.209 -> 0.245 ms at C3).
        s_skip[tid] = out_tiles == 0u;   // (reused: the SH rows are long in registers)
        s_rec[tid * kRecQuads + 0] = q0; s_rec[tid * kRecQuads + 1] = q1; s_rec[tid * kRecQuads + 2] = q2; s_rec[tid * kRecQuads + 3] = q3;
        s_rec[tid * kRecQuads + 4] = q4;
        if (want_jac) {
#pragma unroll
            for (int k = 0; k < 9; ++k) s_jac[tid * 9 + k] = J[k];   // (stride 9 words: conflict-free)
        }
        __syncthreads();
        {
            const int nrows = min(kPreBlock, P - base);
            float4* dst = recs + (size_t)base * kRecQuads;
            for (int fq = tid; fq < nrows * kRecQuads; fq += kPreBlock) {
                const int c0 = fq & ~7, r0 = c0 / kRecQuads, r1 = min((c0 + 7) / kRecQuads, nrows - 1);   // the rows of this chunk's 128-B line (2 or 3)
                if (!(s_skip[r0] & s_skip[min(r0 + 1, r1)] & s_skip[r1])) dst[fq] = s_rec[fq];
            }
            // 9 floats per row: rows * 9 floats are 16-B aligned per block of 128 rows (128 * 36 B) -- whole float4 chunks, plus a scalar tail
            if (want_jac) {
                float* jd = f.sh_jac + (size_t)base * 9;
                const int nf = nrows * 9, nq = nf / 4;
                for (int fq = tid; fq < nq; fq += kPreBlock) {
                    const int f0 = (fq & ~7) * 4, r0 = f0 / 9, r1 = min((f0 + 31) / 9, nrows - 1);       // the rows of this chunk's 128-B line (4 or 5)
                    bool dead = true;
                    for (int r = r0; r <= r1; ++r) dead = dead && s_skip[r];
                    if (!dead) reinterpret_cast<float4*>(jd)[fq] = reinterpret_cast<const float4*>(s_jac)[fq];
                }
                for (int fk = nq * 4 + tid; fk < nf; fk += kPreBlock) jd[fk] = s_jac[fk];
            }
        }
        if (!in_range) return;
        radii[i] = out_radius; tiles_touched[i] = out_tiles; rect[i] = out_rect; depth_keys[i] = out_key; clamped[i] = out_clamped;
        return;
    }
    radii[i] = out_radius;
    tiles_touched[i] = out_tiles;
    rect[i] = out_rect;
    depth_keys[i] = out_key;
    clamped[i] = out_clamped;
    float4* rec = recs + (size_t)i * kRecQuads;   // (the path without SH staging: precomputed colours)
    rec[0] = q0; rec[1] = q1; rec[2] = q2; rec[3] = q3; rec[4] = q4;
}

// ---------------------------------------------------------------------------------------------
// K8: per-Gaussian backward (Appendix A.6).  First sums the Gaussian's per-(tile, Gaussian) gradient records: K7 stores
// them in emission order, where they form the contiguous span [first, first + tiles_touched) (first[] = FrameDev.first, a compact
// array by Gaussian id written by K3); ascending order -> deterministic.  Slots whose `written` flag is clear got no record from K7 (no pixel
// contributed) and are skipped without being read.
// ---------------------------------------------------------------------------------------------
template <bool kLdsSH, int NC>
__global__ __launch_bounds__(kPreBlock) void preprocess_backward_kernel(
    int P, FrameDev f, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ transMat_precomp,
    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped, const float4* __restrict__ recs,
    const float4* __restrict__ inst_grads, const uint8_t* __restrict__ written, const uint32_t* __restrict__ tiles_touched,
    SrGradients out) {
    __shared__ __attribute__((aligned(16))) float s_sh[kLdsSH ? kPreBlock * kShLdsStride : 4];
    const int tid = threadIdx.x, base = blockIdx.x * kPreBlock;
    const int i = base + tid;
    const int M = f.sh_coeffs;
    if (i < P) {
        float g_means3D[3] = {0, 0, 0}, g_scales[2] = {0, 0}, g_rot[4] = {0, 0, 0, 0}, g_m2d[2] = {0, 0};
        float dT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dTr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g_col[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g_opa = 0.f;
        const bool vis = radii[i] > 0 && !(f.overflow && *f.overflow);   // (capacity mode, overflowed frame: nothing was rendered)
        float* dsh_g = (!kLdsSH && out.dL_dsh) ? out.dL_dsh + (size_t)i * M * 3 : nullptr;
        float* row = s_sh + tid * kShLdsStride;
        if (vis) {
            // the splat's transform, centre and (activated) opacity: recomputed from the inputs with K1's own functions -- the same bits as
            // the record holds (asserted by the parity tests: every gradient is unchanged), without fetching 80 B per visible Gaussian
            const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
            // (the backward's arguments do not include the opacities -- the reference's rasterize_gaussians_backward has none: the activated
            // value is the one float still taken from the record, and only where it is needed: the sigmoid adjoint, or a TIGHTBBOX build)
            float act_opacity = 1.f;
            if (SR_TIGHTBBOX || (f.activations & SR_ACT_SIGMOID_OPACITY)) act_opacity = reinterpret_cast<const float*>(recs)[(size_t)i * kRecFloats + 11];
            float Tm[9];
            float qn = 1.f;
            float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
            float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
            float2 sc = make_float2(1.f, 1.f);
            if (transMat_precomp) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Tm[k] = transMat_precomp[9 * (size_t)i + k];
            } else {
                float Bf[12];
                build_B(f.proj, f.W, f.H, Bf);   // (the forward's viewport: K1's transMat)
                q = load_rotation(rotations, i, f.activations, qn);
                quat_to_R(q, R);
                sc = load_scales(scales, i, f.activations);
                splat_transform(Bf, px, py, pz, R, f.scale_modifier * sc.x, f.scale_modifier * sc.y, Tm);
            }
            const SplatBox box = splat_box(Tm, bbox_cutoff(act_opacity));
            float4 g0, g1, g2, g3, g4, g5;
            {
                constexpr int kGQ = NC == 9 ? kGradQuads + 1 : kGradQuads;   // quads per gradient record
                float4 g[kGQ];
#pragma unroll
                for (int k = 0; k < kGQ; ++k) g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                const uint32_t first = first_index(f, (uint32_t)i), cnt = tiles_touched[i];
                // Four records per trip.  K7 writes a record only where some pixel contributed -- about 40 % of a Gaussian's span --
                // and sets the slot's byte in `written`; the flags of the next trip are in flight behind this trip's records.
                constexpr int kTrip = 4;
                const uint32_t end = first + cnt;
                auto load_flags = [&](uint32_t e, uint8_t (&fl)[kTrip]) {
#pragma unroll
                    for (int t = 0; t < kTrip; ++t) fl[t] = e + t < end ? written[e + t] : (uint8_t)0;
                };
                uint8_t fl[kTrip], fn[kTrip] = {0, 0, 0, 0};
                load_flags(first, fl);
                for (uint32_t e = first; e < end; e += kTrip) {
                    if (e + kTrip < end) load_flags(e + kTrip, fn);
                    float4 a[kTrip][kGQ];
#pragma unroll
                    for (int t = 0; t < kTrip; ++t) {
                        const float4* gr = inst_grads + (size_t)(e + t) * kGQ;
#pragma unroll
                        for (int k = 0; k < kGQ; ++k) a[t][k] = fl[t] ? gr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int t = 0; t < kTrip; ++t) {
                        if (fl[t]) {
#pragma unroll
                            for (int k = 0; k < kGQ; ++k) { g[k].x += a[t][k].x; g[k].y += a[t][k].y; g[k].z += a[t][k].z; g[k].w += a[t][k].w; }
                        }
                        fl[t] = fn[t];
                    }
                }
                g0 = g[0]; g1 = g[1]; g2 = g[2]; g3 = g[3]; g4 = g[4]; g5 = g[5];
                if (NC == 9) { g_col[6] = g[kGQ - 1].x; g_col[7] = g[kGQ - 1].y; g_col[8] = g[kGQ - 1].z; }
            }
            const float Tu[3] = {Tm[0], Tm[1], Tm[2]}, Tv[3] = {Tm[3], Tm[4], Tm[5]}, Tw[3] = {Tm[6], Tm[7], Tm[8]};
            // moments -> dL/dT (see common.h).  The records carry the moments about the Gaussian's own centre c = (cx, cy) (K7's flush), i.e.
            // of the same ray-splat form written in x' = x - cx, y' = y - cy with Tu' = Tu - cx Tw, Tv' = Tv - cy Tw:
            //   dTu' = Tv' x S0 - Tw x Sy',  dTv' = S0 x Tu' - Sx' x Tw,  dTw' = Tu' x Sy' - Tv' x Sx' + Z
            // and back through the shift: dTu = dTu', dTv = dTv', dTw = dTw' - cx dTu' - cy dTv'.  Every cross product now multiplies
            // quantities of the splat's own extent instead of pixel coordinates ~1000 that cancel.
            {
                // (the reference point is the centre CLAMPED INTO THE IMAGE, like K7's flush: a splat whose centre projects thousands of pixels
                // off-screen would otherwise cancel lever arms of that length)
                const float cx = fminf(fmaxf(box.cx, 0.f), (float)(f.W - 1)), cy = fminf(fmaxf(box.cy, 0.f), (float)(f.H - 1));
                const float Tuc[3] = {Tu[0] - cx * Tw[0], Tu[1] - cx * Tw[1], Tu[2] - cx * Tw[2]};
                const float Tvc[3] = {Tv[0] - cy * Tw[0], Tv[1] - cy * Tw[1], Tv[2] - cy * Tw[2]};
                const float S0[3] = {g0.x, g0.y, g0.z}, Sx[3] = {g0.w, g1.x, g1.y}, Sy[3] = {g1.z, g1.w, g2.x}, Z[3] = {g2.y, g2.z, g2.w};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                    dT[0 + c] = (Tvc[c1] * S0[c2] - Tvc[c2] * S0[c1]) - (Tw[c1] * Sy[c2] - Tw[c2] * Sy[c1]);
                    dT[3 + c] = (S0[c1] * Tuc[c2] - S0[c2] * Tuc[c1]) - (Sx[c1] * Tw[c2] - Sx[c2] * Tw[c1]);
                    dT[6 + c] = (Tuc[c1] * Sy[c2] - Tuc[c2] * Sy[c1]) - (Tvc[c1] * Sx[c2] - Tvc[c2] * Sx[c1]) + Z[c];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) dT[6 + c] = (dT[6 + c] - cx * dT[0 + c]) - cy * dT[3 + c];
            }
            const float gx2 = g3.x, gy2 = g3.y;
            g_opa = g3.z;
            if (f.activations & SR_ACT_SIGMOID_OPACITY) { const float o = act_opacity; g_opa *= o * (1.f - o); }
            const float gn[3] = {g3.w, g4.x, g4.y};
            g_col[0] = g4.z; g_col[1] = g4.w; g_col[2] = g5.x;
            if (NC >= 6) { g_col[3] = g5.y; g_col[4] = g5.z; g_col[5] = g5.w; }
            // densification proxy from the blend-only dL/dT (Appendix A.6, last paragraph)
#if SR_PROXY_DEPTH_VIEW_Z
            const float depth_c = ((f.view[2] * px + f.view[6] * py) + f.view[10] * pz) + f.view[14];   // view-space depth of the centre, as K1 computes it
#else
            const float depth_c = Tw[2];      // upstream: transMat[8]
#endif
            g_m2d[0] = dT[2] * depth_c * 0.5f * (float)f.bw_W;
            g_m2d[1] = dT[5] * depth_c * 0.5f * (float)f.bw_H;
#pragma unroll
            for (int k = 0; k < 9; ++k) dTr[k] = dT[k];
            if (gx2 != 0.f || gy2 != 0.f) {
                const float t[3] = {9.f, 9.f, -1.f};
                const float d = (t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1]) + t[2] * Tw[2] * Tw[2];
                float fv[3], dLdd = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) { fv[c] = t[c] / d; dLdd += (gx2 * Tu[c] * Tw[c] + gy2 * Tv[c] * Tw[c]) * fv[c]; }
                dLdd *= (-1.f / d);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dT[0 + c] += gx2 * fv[c] * Tw[c];
                    dT[3 + c] += gy2 * fv[c] * Tw[c];
                    dT[6 + c] += gx2 * fv[c] * Tu[c] + gy2 * fv[c] * Tv[c] + dLdd * (t[c] * Tw[c] * 2.f);
                }
            }
            if (!transMat_precomp) {
                float B[12];
                build_B(f.proj, f.bw_W, f.bw_H, B);   // (SR_BACKWARD_WH_FROM_FOCAL: upstream's backward derives the size from focal * tanfov)
                // q, R, sc from above; the chain below uses the scales WITHOUT the modifier (upstream quirk, A.6)
                float dL0[3], dL1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dL0[k] = B[0 + k] * dT[0] + B[4 + k] * dT[3] + B[8 + k] * dT[6];
                    dL1[k] = B[0 + k] * dT[1] + B[4 + k] * dT[4] + B[8 + k] * dT[7];
                    g_means3D[k] = B[0 + k] * dT[2] + B[4 + k] * dT[5] + B[8 + k] * dT[8];
                }
                const float* v = f.view;
                float dtn[3] = {v[0] * gn[0] + v[1] * gn[1] + v[2] * gn[2], v[4] * gn[0] + v[5] * gn[1] + v[6] * gn[2],
                                v[8] * gn[0] + v[9] * gn[1] + v[10] * gn[2]};
                const float vx = ((v[0] * px + v[4] * py) + v[8] * pz) + v[12];
                const float vy = ((v[1] * px + v[5] * py) + v[9] * pz) + v[13];
                const float vz = ((v[2] * px + v[6] * py) + v[10] * pz) + v[14];
                const float nx = R[2], ny = R[5], nz = R[8];
                const float n0 = (v[0] * nx + v[4] * ny) + v[8] * nz;
                const float n1 = (v[1] * nx + v[5] * ny) + v[9] * nz;
                const float n2 = (v[2] * nx + v[6] * ny) + v[10] * nz;
                const float cosv = -((vx * n0 + vy * n1) + vz * n2);
                const float mult = cosv > 0.f ? 1.f : -1.f;
                dtn[0] *= mult; dtn[1] *= mult; dtn[2] *= mult;
                float G[9];
#pragma unroll
                for (int k = 0; k < 3; ++k) { G[3 * k + 0] = dL0[k] * sc.x; G[3 * k + 1] = dL1[k] * sc.y; G[3 * k + 2] = dtn[k]; }
                g_scales[0] = dL0[0] * R[0] + dL0[1] * R[3] + dL0[2] * R[6];
                g_scales[1] = dL1[0] * R[1] + dL1[1] * R[4] + dL1[2] * R[7];
                const float r = q.x, x = q.y, y = q.z, z = q.w;
                g_rot[0] = 2.f * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
                g_rot[1] = 2.f * (y * G[1] + z * G[2] + y * G[3] - 2.f * x * G[4] - r * G[5] + z * G[6] + r * G[7] - 2.f * x * G[8]);
                g_rot[2] = 2.f * (-2.f * y * G[0] + x * G[1] + r * G[2] + x * G[3] + z * G[5] - r * G[6] + z * G[7] - 2.f * y * G[8]);
                g_rot[3] = 2.f * (-2.f * z * G[0] - r * G[1] + x * G[2] + r * G[3] - 2.f * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
                // adjoints of the fused activations (raw parameters in, raw gradients out)
                if (f.activations & SR_ACT_EXP_SCALES) { g_scales[0] *= sc.x; g_scales[1] *= sc.y; }
                if (f.activations & SR_ACT_NORMALIZE_ROTATIONS) {
                    const float qg = ((q.x * g_rot[0] + q.y * g_rot[1]) + q.z * g_rot[2]) + q.w * g_rot[3];
                    g_rot[0] = (g_rot[0] - q.x * qg) / qn; g_rot[1] = (g_rot[1] - q.y * qg) / qn;
                    g_rot[2] = (g_rot[2] - q.z * qg) / qn; g_rot[3] = (g_rot[3] - q.w * qg) / qn;
                }
            }
            if (shs) {
                const float ox = px - f.campos[0], oy = py - f.campos[1], oz = pz - f.campos[2];
                const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
                const float x = ox / len, y = oy / len, z = oz / len;
                const uint8_t cl = clamped[i];
                const float gc[3] = {(cl & 1) ? 0.f : g_col[0], (cl & 2) ? 0.f : g_col[1], (cl & 4) ? 0.f : g_col[2]};
                // with SHs as the colour source dL_dcolors (optional) carries the clamp-masked dL/drgb: the one per-view
                // input of the SH adjoint, which is all a frame-parallel rank has to ship (sh_gradient_expand_kernel)
                g_col[0] = gc[0]; g_col[1] = gc[1]; g_col[2] = gc[2];
                if (kLdsSH) { if (out.dL_dsh) sh_basis_adjoint(f.sh_degree, M, row, x, y, z, gc); }
                else if (dsh_g) sh_basis_adjoint(f.sh_degree, M, dsh_g, x, y, z, gc);
                // the colour's dependence on the centre through the view direction: K1 left d rgb / d centre (9 floats) behind
                const float* J = f.sh_jac + 9 * (size_t)i;
#pragma unroll
                for (int k = 0; k < 3; ++k) g_means3D[k] += (gc[0] * J[k] + gc[1] * J[3 + k]) + gc[2] * J[6 + k];
            }
        } else if (shs) {
            if (kLdsSH) {
#pragma unroll
                for (int k = 0; k < kShRowFloats / 4; ++k) reinterpret_cast<float4*>(row)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (dsh_g) {
                for (int k = 0; k < M * 3; ++k) dsh_g[k] = 0.f;
            }
        }
        // (sending these two 12-B-per-Gaussian outputs through LDS behind the dL_dsh rows, as K1 does with its records, changes nothing here:
        // 0.345 vs 0.347 ms -- 72 MB of K8's 1.8 GB)
        if (out.dL_dmeans3D) { out.dL_dmeans3D[3 * (size_t)i] = g_means3D[0]; out.dL_dmeans3D[3 * (size_t)i + 1] = g_means3D[1]; out.dL_dmeans3D[3 * (size_t)i + 2] = g_means3D[2]; }
        if (out.dL_dmeans2D) { out.dL_dmeans2D[3 * (size_t)i] = g_m2d[0]; out.dL_dmeans2D[3 * (size_t)i + 1] = g_m2d[1]; out.dL_dmeans2D[3 * (size_t)i + 2] = 0.f; }
        if (out.dL_dscales) {
            if ((reinterpret_cast<uintptr_t>(out.dL_dscales) & 7u) == 0) reinterpret_cast<float2*>(out.dL_dscales)[i] = make_float2(g_scales[0], g_scales[1]);
            else { out.dL_dscales[2 * (size_t)i] = g_scales[0]; out.dL_dscales[2 * (size_t)i + 1] = g_scales[1]; }
        }
        if (out.dL_drotations) { reinterpret_cast<float4*>(out.dL_drotations)[i] = make_float4(g_rot[0], g_rot[1], g_rot[2], g_rot[3]); }
        if (out.dL_dopacity) out.dL_dopacity[i] = g_opa;
        if (out.dL_dcolors) {
            if (NC == 9) {   // the six precomputed channels; rgb went through the SH adjoint
#pragma unroll
                for (int c = 0; c < 6; ++c) out.dL_dcolors[6 * (size_t)i + c] = g_col[3 + c];
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) out.dL_dcolors[NC * (size_t)i + c] = g_col[c];
            }
        }
        if (out.dL_dtransMat) {
            // upstream writes the AABB-centre-augmented dL/dT back only when transMat is an input (A.6)
#pragma unroll
            for (int k = 0; k < 9; ++k) out.dL_dtransMat[9 * (size_t)i + k] = transMat_precomp ? dT[k] : dTr[k];
        }
    }
    if (kLdsSH && out.dL_dsh) {
        __syncthreads();
        sh_rows_from_lds(out.dL_dsh, base, P, s_sh, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// Colour gradients only (sr_backward_colors): sums slots 18..20 of a Gaussian's written gradient records -- the same records,
// flags and ascending order as K8, so the sums are the ones K8 forms -- and applies the SH clamp mask.  This is all a
// frame-parallel rank has to ship for the factored SH exchange (12 B per Gaussian), and it is available right after K7: the
// all-gather can run while K8 does the rest of the backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void color_gradient_kernel(int P, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                             const float4* __restrict__ recs, const float4* __restrict__ inst_grads,
                                                             const uint8_t* __restrict__ written, const uint32_t* __restrict__ tiles_touched,
                                                             FrameDev f, int mask_clamped, float* __restrict__ dL_dcolors) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (radii[i] > 0 && !(f.overflow && *f.overflow)) {
        const uint32_t first = first_index(f, (uint32_t)i), end = first + tiles_touched[i];
        // four slots per trip, the next trip's flags in flight behind this trip's records (as in K8; same ascending order of the adds)
        constexpr int kTrip = 4;
        auto load_flags = [&](uint32_t e, uint8_t (&fl)[kTrip]) {
#pragma unroll
            for (int t = 0; t < kTrip; ++t) fl[t] = e + t < end ? written[e + t] : (uint8_t)0;
        };
        uint8_t fl[kTrip], fn[kTrip] = {0, 0, 0, 0};
        load_flags(first, fl);
        for (uint32_t e = first; e < end; e += kTrip) {
            if (e + kTrip < end) load_flags(e + kTrip, fn);
            float4 a4[kTrip]; float a5[kTrip];
#pragma unroll
            for (int t = 0; t < kTrip; ++t) {
                const float4* gr = inst_grads + (size_t)(e + t) * kGradQuads;
                a4[t] = fl[t] ? gr[4] : make_float4(0.f, 0.f, 0.f, 0.f);
                a5[t] = fl[t] ? gr[5].x : 0.f;
            }
#pragma unroll
            for (int t = 0; t < kTrip; ++t) {
                if (fl[t]) { c0 += a4[t].z; c1 += a4[t].w; c2 += a5[t]; }
                fl[t] = fn[t];
            }
        }
        if (mask_clamped) { const uint8_t cl = clamped[i]; if (cl & 1) c0 = 0.f; if (cl & 2) c1 = 0.f; if (cl & 4) c2 = 0.f; }
    }
    dL_dcolors[3 * (size_t)i] = c0; dL_dcolors[3 * (size_t)i + 1] = c1; dL_dcolors[3 * (size_t)i + 2] = c2;
}

// ---------------------------------------------------------------------------------------------
// SH gradient of V frames of the same Gaussians from their clamp-masked colour gradients:
//   dL_dsh[i][k][c] = sum_v basis_k(dir(p_i, campos_v)) * gc[v][i][c]
// The SH adjoint is linear in gc and its only other per-view input is the camera position, so frame-parallel ranks
// all-gather 12 B per Gaussian instead of all-reducing the 192-B dL_dsh rows and every rank expands the sum locally.
// b[k] * g is evaluated exactly as sh_backward does, so V = 1 reproduces K8's dL_dsh bit for bit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float b[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = 0.f;
    b[0] = kSH_C0;
    if (deg > 0) {
        b[1] = -kSH_C1 * y; b[2] = kSH_C1 * z; b[3] = -kSH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = kSH_C2[0] * xy; b[5] = kSH_C2[1] * yz; b[6] = kSH_C2[2] * (2.f * zz - xx - yy);
            b[7] = kSH_C2[3] * xz; b[8] = kSH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = kSH_C3[0] * y * (3.f * xx - yy);
                b[10] = kSH_C3[1] * xy * z;
                b[11] = kSH_C3[2] * y * (4.f * zz - xx - yy);
                b[12] = kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = kSH_C3[4] * x * (4.f * zz - xx - yy);
                b[14] = kSH_C3[5] * z * (xx - yy);
                b[15] = kSH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

template <bool kLdsSH>
__global__ __launch_bounds__(kPreBlock) void sh_gradient_expand_kernel(int P, int M, int deg, int V, const float* __restrict__ means3D,
                                                                  const float* __restrict__ campos, const float* __restrict__ gc,
                                                                  float* __restrict__ dL_dsh) {
    __shared__ __attribute__((aligned(16))) float s_sh[kLdsSH ? kPreBlock * kShLdsStride : 4];
    const int tid = threadIdx.x, base = blockIdx.x * kPreBlock;
    const int i = base + tid;
    if (i < P) {
        float acc[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) acc[k] = 0.f;
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
        for (int v = 0; v < V; ++v) {
            const float* g = gc + ((size_t)v * P + i) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;   // not visible in this view (or fully clamped): adds +0
            const float ox = px - campos[3 * v], oy = py - campos[3 * v + 1], oz = pz - campos[3 * v + 2];
            const float len = sqrtf((ox * ox + oy * oy) + oz * oz);
            float b[16];
            sh_basis(deg, ox / len, oy / len, oz / len, b);
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc[3 * k] += b[k] * g0; acc[3 * k + 1] += b[k] * g1; acc[3 * k + 2] += b[k] * g2; }
        }
        if (kLdsSH) {
            float4* row = reinterpret_cast<float4*>(s_sh + tid * kShLdsStride);
#pragma unroll
            for (int k = 0; k < kShRowFloats / 4; ++k) row[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
        } else {
            float* row = dL_dsh + (size_t)i * M * 3;
#pragma unroll
            for (int k = 0; k < 48; ++k) { if (k < M * 3) row[k] = acc[k]; }   // constant indices: acc stays in registers
            for (int k = 48; k < M * 3; ++k) row[k] = 0.f;
        }
    }
    if (kLdsSH) {
        __syncthreads();
        sh_rows_from_lds(dL_dsh, base, P, s_sh, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// K9
// ---------------------------------------------------------------------------------------------
__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float vz = ((view[2] * means3D[3 * i] + view[6] * means3D[3 * i + 1]) + view[10] * means3D[3 * i + 2]) + view[14];
    present[i] = vz > kNear ? 1 : 0;
}

// host launchers ---------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

hipError_t launch_preprocess_forward(int P, const FrameDev& f, const SrGaussians& g, float4* recs, uint32_t* depth_keys,
                                     uint32_t* tiles_touched, uint2* rect, uint8_t* clamped, int32_t* radii, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const dim3 grid((P + kPreBlock - 1) / kPreBlock), block(kPreBlock);
    if (g.shs && g.sh_coeffs == 16 && aligned16(g.shs))
        hipLaunchKernelGGL(preprocess_forward_kernel<true>, grid, block, 0, s, P, f, g.means3D, g.opacities, g.scales,
                           g.rotations, g.shs, g.colors_precomp, g.transMat_precomp, g.mask, recs, depth_keys, tiles_touched, rect, clamped, radii);
    else
        hipLaunchKernelGGL(preprocess_forward_kernel<false>, grid, block, 0, s, P, f, g.means3D, g.opacities, g.scales,
                           g.rotations, g.shs, g.colors_precomp, g.transMat_precomp, g.mask, recs, depth_keys, tiles_touched, rect, clamped, radii);
    return hipGetLastError();
}

hipError_t launch_preprocess_backward(int P, const FrameDev& f, const SrGaussians& g, const int32_t* radii,
                                      const uint8_t* clamped, const float4* recs, const float4* inst_grads, const uint8_t* written,
                                      const uint32_t* tiles_touched, const SrGradients& out, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const dim3 grid((P + kPreBlock - 1) / kPreBlock), block(kPreBlock);
    if (f.colors == 9 && g.sh_coeffs == 16 && aligned16(g.shs) && (!out.dL_dsh || aligned16(out.dL_dsh)))
        hipLaunchKernelGGL((preprocess_backward_kernel<true, 9>), grid, block, 0, s, P, f, g.means3D, g.scales, g.rotations, g.shs,
                           g.transMat_precomp, radii, clamped, recs, inst_grads, written, tiles_touched, out);
    else if (f.colors == 9)
        hipLaunchKernelGGL((preprocess_backward_kernel<false, 9>), grid, block, 0, s, P, f, g.means3D, g.scales, g.rotations, g.shs,
                           g.transMat_precomp, radii, clamped, recs, inst_grads, written, tiles_touched, out);
    else if (f.colors == 6)
        hipLaunchKernelGGL((preprocess_backward_kernel<false, 6>), grid, block, 0, s, P, f, g.means3D, g.scales, g.rotations, g.shs,
                           g.transMat_precomp, radii, clamped, recs, inst_grads, written, tiles_touched, out);
    else if (g.shs && g.sh_coeffs == 16 && aligned16(g.shs) && (!out.dL_dsh || aligned16(out.dL_dsh)))
        hipLaunchKernelGGL((preprocess_backward_kernel<true, 3>), grid, block, 0, s, P, f, g.means3D, g.scales, g.rotations, g.shs,
                           g.transMat_precomp, radii, clamped, recs, inst_grads, written, tiles_touched, out);
    else
        hipLaunchKernelGGL((preprocess_backward_kernel<false, 3>), grid, block, 0, s, P, f, g.means3D, g.scales, g.rotations, g.shs,
                           g.transMat_precomp, radii, clamped, recs, inst_grads, written, tiles_touched, out);
    return hipGetLastError();
}

hipError_t launch_sh_gradient_expand(int P, int M, int deg, int V, const float* means3D, const float* campos, const float* gc,
                                     float* dL_dsh, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const dim3 grid((P + kPreBlock - 1) / kPreBlock), block(kPreBlock);
    if (M == 16 && aligned16(dL_dsh))
        hipLaunchKernelGGL(sh_gradient_expand_kernel<true>, grid, block, 0, s, P, M, deg, V, means3D, campos, gc, dL_dsh);
    else
        hipLaunchKernelGGL(sh_gradient_expand_kernel<false>, grid, block, 0, s, P, M, deg, V, means3D, campos, gc, dL_dsh);
    return hipGetLastError();
}

hipError_t launch_color_gradients(int P, const FrameDev& f, const int32_t* radii, const uint8_t* clamped, const float4* recs, const float4* inst_grads,
                                  const uint8_t* written, const uint32_t* tiles_touched, bool mask_clamped, float* dL_dcolors, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(color_gradient_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, clamped, recs, inst_grads, written, tiles_touched,
                       f, mask_clamped ? 1 : 0, dL_dcolors);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    return hipGetLastError();
}

}  // namespace sr
