// Micro-benchmark (round 6): the per-entry 21-value wave reduction of K7 through the LDS instead of through DPP / lane swaps.
//   hipcc --offload-arch=gfx950 -O3 -I streetunveiler_amd/csrc -I include tools/ubench/lds_reduce_ubench.hip -o /tmp/lds_reduce_ubench && /tmp/lds_reduce_ubench
// Every lane stores its 21 partial sums into a [value][lane] array (rows padded by four floats: the skew that makes the readers' b128 loads
// conflict-free); reader lane (value k, half t) adds the 32 partials of its half with eight ds_read_b128 + 32 v_add, one quad-perm DPP add joins
// the halves.  The vector unit issues 21 + 33 instructions per entry instead of 21 + ~60 (36 of them DPP at 1.8 ns); the stores and loads go
// down the LDS pipe, which K7 leaves mostly idle.  What is measured: whole-loop time per entry with a filler that stands in for the
// quadrant tests (two blocks of ~85 vector instructions, six broadcast ds_read_b128 of the staged entry), three waves per SIMD (the LDS
// allocation caps the occupancy as K7's registers do).
// MODE 0: wave_reduce24<21> (K7 today) | 1: LDS, ds_write_b32 | 2: LDS, ds_write_addtid_b32 | 3: one DPP level (lane ^ 8) first, then 12 values
// through the LDS | 4: no reduction (the filler alone + 21 adds that keep it alive)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "blend_common.h"
using namespace sr;

typedef float f4 __attribute__((ext_vector_type(4)));
// an LDS quad load the compiler cannot split (it turned the float4 loads of the padded rows into ds_read2_b32 pairs: 6-way bank conflicts)
__device__ __forceinline__ f4 lds_read_b128(uint32_t byte_addr) { f4 x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(byte_addr)); return x; }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p; }
template <int kPat> __device__ __forceinline__ float swz(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), kPat)); }
// dpp_fold_rows<21> with the partner's value fetched by ds_swizzle (LDS crossbar, no vector instruction) and a plain add: 2 x 1.05 ns per pair of
// registers instead of 2 x 1.8 ns of DPP adds.  Same value layout afterwards (v[0..5]: value k + 6 bit2 + 12 bit3 over {l, l^4, l^8, l^12}).
__device__ __forceinline__ void swz_fold_rows21(float (&v)[24], int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    float t[12], u[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { t[k] = swz<0x201F>(v[k]); u[k] = (k + 12 < 21) ? swz<0x201F>(v[k + 12]) : 0.f; }   // lane ^ 8
    // (the second add of a pair runs under the exec mask of the lanes that keep the other register: a real branch -- the empty asm keeps the
    // compiler from turning it into 21 selects)
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] += t[k];
    if (b3) {
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = v[k + 12] + u[k];
        v[9] = 0.f; v[10] = 0.f; v[11] = 0.f;
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { t[k] = swz<0x101F>(v[k]); u[k] = swz<0x101F>(v[k + 6]); }                           // lane ^ 4
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += t[k];
    if (b2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = v[k + 6] + u[k];
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    }
}
__device__ __forceinline__ float wave_reduce24_swz(float (&v)[24], int lane) {
    swz_fold_rows21(v, lane);
#pragma unroll
    for (int k = 0; k < 3; ++k) fold32(v[k], v[k + 3]);
    float z = 0.f;
    fold16(v[0], v[1]); fold16(v[2], z);
    float a = v[0], b = v[2];
    a += dpp_mov<0x4E>(a); b += dpp_mov<0x4E>(b);
    a += dpp_mov<0xB1>(a); b += dpp_mov<0xB1>(b);
    return (lane & 2) ? b : a;
}
constexpr int kRow = 68;   // floats per row of the transposition array

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k(float* __restrict__ out, const float4* __restrict__ ein, int iters, int fill) {
    __shared__ __attribute__((aligned(16))) float s_raw[13 * 256];   // 13 KB per workgroup: 12 workgroups per CU = three waves per SIMD
    float4 (*s_e)[64] = reinterpret_cast<float4 (*)[64]>(s_raw);                    // 6 KB: the staged entries
    float* s_t = s_raw + 6 * 256;                                                    // 24 x kRow floats: the transposition array (MODE 1..3)
    float (*s_out)[24] = reinterpret_cast<float (*)[24]>(s_raw + 6 * 256);           // 6 KB: the round's records (MODE 0)
    float* s_pad = s_raw + 13 * 256 - 64;
    const int lane = threadIdx.x;
    for (int q = 0; q < 6; ++q) s_e[q][lane] = ein[q * 64 + lane];
    s_pad[lane] = 0.f;
    for (int i = lane; i < 24 * kRow; i += 64) s_t[i] = 0.f;
    __syncthreads();
    // reader mapping: the ds_read_b128 lane groups of gfx950 are {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): position p inside the group
    const int l5 = lane & 31;
    int g, p;
    if (l5 < 4) { g = 0; p = l5; } else if (l5 < 12) { g = 1; p = l5 - 4; } else if (l5 < 16) { g = 0; p = l5 - 8; }
    else if (l5 < 20) { g = 1; p = l5 - 8; } else if (l5 < 28) { g = 0; p = l5 - 12; } else { g = 1; p = l5 - 16; }
    g += (lane >> 5) * 2;
    const int rt = p & 1;
    const int rk = g * 8 + (p >> 1);             // value index of this reader (MODE 1, 2): 0..31
    const float* rrow = s_t + (rk < 24 ? rk : 0) * kRow + rt * 32;
    // MODE 3: after the lane ^ 8 fold a lane with bit 3 clear holds values 0..11 (+ nothing), with bit 3 set values 12..23; 32 partials per value in the
    // lanes with that bit-3 value -> the array is [24][32 (+ pad)]: lane l writes slot ((l >> 4) << 3) | (l & 7) of rows k (+12)
    const int w3slot = ((lane >> 4) << 3) | (lane & 7), w3base = (lane & 8) ? 12 : 0;
    const float* rrow3 = s_t + (rk < 24 ? rk : 0) * 40 + rt * 4;   // rows of 40 floats, reader half t takes the quads 2 i + t: bank sets (2 u + t) distinct inside a b128 lane group
    const bool holds_total = reduce24_holds_total(lane) && reduce24_index(lane) < 21;
    float acc = 0.f;
    const float xl = (float)(lane & 7) - 8.f, yl = (float)(lane >> 3) - 8.f;
    for (int it = 0; it < iters; ++it) {
        const int j = it & 63;
        const float4 e0 = s_e[0][j], e1 = s_e[1][j], e2 = s_e[2][j], e3 = s_e[3][j], e4 = s_e[4][j], e5 = s_e[5][j];
        float v[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) { v[i] = 0.f; if (i < 21) asm volatile("" : "+v"(v[i])); }
        for (int q = 0; q < fill; ++q) {   // stand-in for a quadrant test: ~85 vector instructions
            const float xq = xl + (float)(8 * (q & 1)), yq = yl + (float)(4 * (q & 2));
            const float ppx = fmaf(xq, e0.x, fmaf(yq, e0.w, e1.z)), ppy = fmaf(xq, e0.y, fmaf(yq, e1.x, e1.w)), ppz = fmaf(xq, e0.z, fmaf(yq, e1.y, e2.x));
            const float pi = fast_rcp(ppz), sx = ppx * pi, sy = ppy * pi;
            const float rho = fminf(sx * sx + sy * sy, 2.f * ((e3.x - xq) * (e3.x - xq) + (e3.y - yq) * (e3.y - yq)));
            const float G = __builtin_amdgcn_exp2f(rho * -0.72f), alpha = fminf(0.99f, e3.z * G);
            const float om = fast_rcp(1.f - alpha), T = om + acc, w = alpha * T;
            const float depth = fmaf(sx, e2.y, fmaf(sy, e2.z, e2.w)), idp = fast_rcp(depth);
            float phi = fmaf(e4.w, xq, fmaf(e5.x, yq, fmaf(e5.y, sx, fmaf(depth, sy, fmaf(e4.x, G, fmaf(e4.y, w, e4.z * T))))));
            const float md = fmaf(idp, -0.2f, 1.f), t1 = fmaf(md, xq, -yq), psi = phi + fmaf(md, t1 - yq, sx);
            const float da = T * psi - om * acc, dz = fmaf(w, fmaf(t1 * (idp * idp), 0.4f, sy), G), dG = e3.z * da;
            v[18] += w * xq; v[19] += w * yq; v[20] += w * sx; v[15] += w * sy; v[16] += w * G; v[17] += w * T; v[14] += G * da; v[11] += dz;
            const float gG = -dG * G;
            const float dpx = (gG * sx + dz * e2.y) * pi, dpy = (gG * sy + dz * e2.z) * pi, dpz = -(dpx * sx + dpy * sy);
            v[0] += dpx; v[1] += dpy; v[2] += dpz;
            v[3] = fmaf(xq, dpx, v[3]); v[4] = fmaf(xq, dpy, v[4]); v[5] = fmaf(xq, dpz, v[5]);
            v[6] = fmaf(yq, dpx, v[6]); v[7] = fmaf(yq, dpy, v[7]); v[8] = fmaf(yq, dpz, v[8]);
            v[9] = fmaf(dz, sx, v[9]); v[10] = fmaf(dz, sy, v[10]);
            v[12] = fmaf(gG, e3.x - xq, v[12]); v[13] = fmaf(gG, e3.y - yq, v[13]);
        }
        if (MODE == 0 || MODE == 8) {
            const float tot = MODE == 8 ? wave_reduce24_swz(v, lane) : wave_reduce24<21>(v, lane);
            if (holds_total) s_out[j][reduce24_index(lane)] = tot;
            acc += tot * 1e-9f;
        } else if (MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6 || MODE == 7) {
            if (MODE == 6) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 21; ++i) s += v[i];
                acc += s * 1e-9f;
            } else if (MODE == 1 || MODE == 5) {
#pragma unroll
                for (int i = 0; i < 21; ++i) s_t[i * kRow + lane] = v[i];
            } else {
                const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)s_t;   // (LDS addresses are 32-bit; uniform)
                asm volatile(
                    "s_mov_b32 m0, %21\n"
                    "ds_write_addtid_b32 %0 offset:0\n"    "ds_write_addtid_b32 %1 offset:272\n"   "ds_write_addtid_b32 %2 offset:544\n"
                    "ds_write_addtid_b32 %3 offset:816\n"  "ds_write_addtid_b32 %4 offset:1088\n"  "ds_write_addtid_b32 %5 offset:1360\n"
                    "ds_write_addtid_b32 %6 offset:1632\n" "ds_write_addtid_b32 %7 offset:1904\n"  "ds_write_addtid_b32 %8 offset:2176\n"
                    "ds_write_addtid_b32 %9 offset:2448\n" "ds_write_addtid_b32 %10 offset:2720\n" "ds_write_addtid_b32 %11 offset:2992\n"
                    "ds_write_addtid_b32 %12 offset:3264\n" "ds_write_addtid_b32 %13 offset:3536\n" "ds_write_addtid_b32 %14 offset:3808\n"
                    "ds_write_addtid_b32 %15 offset:4080\n" "ds_write_addtid_b32 %16 offset:4352\n" "ds_write_addtid_b32 %17 offset:4624\n"
                    "ds_write_addtid_b32 %18 offset:4896\n" "ds_write_addtid_b32 %19 offset:5168\n" "ds_write_addtid_b32 %20 offset:5440\n"
                    :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
                       "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]),
                       "s"(__builtin_amdgcn_readfirstlane(base))
                    : "memory");
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (rk < 21 && MODE != 5 && MODE != 7) {
                const uint32_t a = lds_addr(rrow);
                f4 x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = lds_read_b128(a + 16 * i);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(x[i])); s0 += x[i].x; s1 += x[i].y; s2 += x[i].z; s3 += x[i].w; }
            }
            float tot = (s0 + s1) + (s2 + s3);
            tot += dpp_mov<0xB1>(tot);   // the other half's sum (lane ^ 1)
            if (rt == 0 && rk < 24) out[(size_t)blockIdx.x * 24 + rk] = tot;
            acc += tot * 1e-9f;
        } else if (MODE == 9 || MODE == 10) {
            // the reduction of the PREVIOUS entry, one entry late: its eight quad loads are issued in front of this entry's 21 stores (one wave's LDS
            // operations execute in order: the loads still see the previous partials), the adds follow the stores
            f4 x[8];
            const uint32_t a = lds_addr(rrow);
            if (rk < 21) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = lds_read_b128(a + 16 * i);
            }
            // the moments' shift applied per lane (six fmac), as the real kernel would have to
            v[3] = fmaf(e5.z, v[0], v[3]); v[4] = fmaf(e5.z, v[1], v[4]); v[5] = fmaf(e5.z, v[2], v[5]);
            v[6] = fmaf(e5.w, v[0], v[6]); v[7] = fmaf(e5.w, v[1], v[7]); v[8] = fmaf(e5.w, v[2], v[8]);
            if (MODE == 9) {
#pragma unroll
                for (int i = 0; i < 21; ++i) s_t[i * kRow + lane] = v[i];
            } else {
                const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)s_t;
                asm volatile(
                    "s_mov_b32 m0, %21\n"
                    "ds_write_addtid_b32 %0 offset:0\n"    "ds_write_addtid_b32 %1 offset:272\n"   "ds_write_addtid_b32 %2 offset:544\n"
                    "ds_write_addtid_b32 %3 offset:816\n"  "ds_write_addtid_b32 %4 offset:1088\n"  "ds_write_addtid_b32 %5 offset:1360\n"
                    "ds_write_addtid_b32 %6 offset:1632\n" "ds_write_addtid_b32 %7 offset:1904\n"  "ds_write_addtid_b32 %8 offset:2176\n"
                    "ds_write_addtid_b32 %9 offset:2448\n" "ds_write_addtid_b32 %10 offset:2720\n" "ds_write_addtid_b32 %11 offset:2992\n"
                    "ds_write_addtid_b32 %12 offset:3264\n" "ds_write_addtid_b32 %13 offset:3536\n" "ds_write_addtid_b32 %14 offset:3808\n"
                    "ds_write_addtid_b32 %15 offset:4080\n" "ds_write_addtid_b32 %16 offset:4352\n" "ds_write_addtid_b32 %17 offset:4624\n"
                    "ds_write_addtid_b32 %18 offset:4896\n" "ds_write_addtid_b32 %19 offset:5168\n" "ds_write_addtid_b32 %20 offset:5440\n"
                    :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
                       "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]),
                       "s"(__builtin_amdgcn_readfirstlane(base))
                    : "memory");
            }
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (rk < 21) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(x[i])); s0 += x[i].x; s1 += x[i].y; s2 += x[i].z; s3 += x[i].w; }
            }
            float tot = (s0 + s1) + (s2 + s3);
            tot += dpp_mov<0xB1>(tot);
            if (rt == 0 && rk < 24) out[(size_t)blockIdx.x * 24 + rk] = tot;
            acc += tot * 1e-9f;
        } else if (MODE == 3) {
            // lane ^ 8 level in registers (24 DPP adds, as dpp_fold_rows' first level), then 12 values per lane through the LDS
            asm volatile(
                "s_nop 1\n"
                "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n" "v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n" "v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                "v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n" "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n" "v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                "v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n" "v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                "v_add_f32_dpp %8, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                "s_nop 1\n"
                : "+&v"(v[0]), "+&v"(v[1]), "+&v"(v[2]), "+&v"(v[3]), "+&v"(v[4]), "+&v"(v[5]), "+&v"(v[6]), "+&v"(v[7]), "+&v"(v[8]), "+&v"(v[9]), "+&v"(v[10]), "+&v"(v[11])
                : "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]));
            // every lane now holds 12 sums over {l, l ^ 8}: values 0..11 in the lanes with bit 3 clear, 12..23 in the others; 32 partials per value
#pragma unroll
            for (int i = 0; i < 12; ++i) s_t[(w3base + i) * 40 + w3slot] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (rk < 24) {
                const uint32_t a = lds_addr(rrow3);
                f4 x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = lds_read_b128(a + 32 * i);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(x[i])); s0 += x[i].x; s1 += x[i].y; s2 += x[i].z; s3 += x[i].w; }
            }
            float tot = (s0 + s1) + (s2 + s3);
            tot += dpp_mov<0xB1>(tot);
            if (rt == 0 && rk < 24) out[(size_t)blockIdx.x * 24 + rk] = tot;
            acc += tot * 1e-9f;
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 21; ++i) s += v[i];
            acc += s * 1e-9f;
        }
    }
    out[(size_t)gridDim.x * 24 + blockIdx.x * 64 + lane] = acc + s_out[lane][lane % 24] + s_pad[lane];
}

template <int MODE> float run(float* d, const float4* e, int iters, int fill) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4 * 3 * 4;  // four batches of three waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, e, 10, fill);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, e, iters, fill); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms;
}
int main(int argc, char** argv) {
    const int blocks = 256 * 4 * 3 * 4;
    float* d; hipMalloc(&d, (size_t)blocks * (24 + 64) * 4);
    float4* e; hipMalloc(&e, 6 * 64 * 16);
    float4 h[6 * 64];
    for (int i = 0; i < 6 * 64; ++i) h[i] = make_float4(0.01f * (i % 7) + 0.1f, 0.02f * (i % 5) + 0.3f, 0.015f * (i % 3) + 1.f, 0.5f + 0.01f * (i % 11));
    hipMemcpy(e, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 1000;
    if (argc > 2) {   // one mode, one fill: for counter passes (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS)
        const int m = atoi(argv[1]), fill = atoi(argv[2]);
        const float t = m == 0 ? run<0>(d, e, iters, fill) : m == 1 ? run<1>(d, e, iters, fill) : m == 2 ? run<2>(d, e, iters, fill) : m == 3 ? run<3>(d, e, iters, fill) : m == 5 ? run<5>(d, e, iters, fill) : m == 6 ? run<6>(d, e, iters, fill) : m == 7 ? run<7>(d, e, iters, fill) : m == 8 ? run<8>(d, e, iters, fill) : m == 9 ? run<9>(d, e, iters, fill) : m == 10 ? run<10>(d, e, iters, fill) : run<4>(d, e, iters, fill);
        printf("mode %d fill %d: %.1f ns per entry and SIMD\n", m, fill, t * 1e6 / (12.0 * iters));
        return 0;
    }
    for (int fill = 0; fill <= 3; ++fill) {
        const float t8 = run<8>(d, e, iters, fill), t9 = run<9>(d, e, iters, fill), t10 = run<10>(d, e, iters, fill);
        const float t5 = run<5>(d, e, iters, fill), t6 = run<6>(d, e, iters, fill), t7 = run<7>(d, e, iters, fill);
        const float t4 = run<4>(d, e, iters, fill), t0 = run<0>(d, e, iters, fill), t1 = run<1>(d, e, iters, fill), t2 = run<2>(d, e, iters, fill), t3 = run<3>(d, e, iters, fill);
        // ns per entry and SIMD: 12 waves per SIMD in total, iters entries each
        auto ns = [&](float ms) { return ms * 1e6 / (12.0 * iters); };
        printf("fill %d: none %.1f | wave_reduce24 %.1f | lds b32 %.1f | lds addtid %.1f | dpp level + lds %.1f   (ns per entry and SIMD, three waves per SIMD)\n",
               fill, ns(t4), ns(t0), ns(t1), ns(t2), ns(t3));
        printf("        b32 writes only %.1f | b128 reads only %.1f | addtid writes only %.1f\n", ns(t5), ns(t6), ns(t7));
        printf("        ds_swizzle folds + adds instead of DPP adds %.1f | LDS, one entry late (loads in front of the stores): b32 %.1f, addtid %.1f\n", ns(t8), ns(t9), ns(t10));
    }
    return 0;
}
