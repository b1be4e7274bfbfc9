// postprocess.hip -- fused allmap post-processing of the render operator:
//   allmap[7,H,W] -> rend_normal (world), surf_depth, surf_point, surf_normal (pseudo-normals)
// i.e. /root/reference/gaussian_renderer/__init__.py:152-177 + /root/reference/utils/point_utils.py:9-37, which the
// reference runs as ~25 full-image torch kernels (plus a 25 MB host->device upload of the pixel grid per call).
// One streaming pass forward, two backward; pure HBM-bound elementwise/stencil work (~60 B/pixel in, ~40 B/pixel out).
//
//   rend_normal = R_c2w * allmap[2:5]
//   expected    = nan_to_num(allmap[0] / allmap[1], 0, 0);  median = nan_to_num(allmap[5], 0, 0)
//   surf_depth  = expected * (1 - depth_ratio) + depth_ratio * median
//   surf_point  = surf_depth * ray_dir(x, y) + cam_pos,  ray_dir = R_c2w * ((x - W/2)/fx, (y - H/2)/fy, 1)
//   surf_normal = normalize((P[y+1,x] - P[y-1,x]) x (P[y,x+1] - P[y,x-1])) * alpha      (0 on the border; alpha detached)
#include "common.h"

namespace sr {

struct PostCam {
    int W, H;
    float fx, fy, depth_ratio;
    const float* view;   // device [16] world_view_transform (W2C^T, row-major)
};

// c2w rotation (row-major R[9]) and position from W2C^T: general 3x3 inverse via the adjugate (the reference calls inverse())
__device__ __forceinline__ void cam_to_world(const float* __restrict__ v, float R[9], float o[3]) {
    // W2C[r][c] = v[4c + r]
    const float a = v[0], b = v[4], c = v[8], d = v[1], e = v[5], f = v[9], g = v[2], h = v[6], i = v[10];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    const float id = 1.f / det;
    R[0] = A * id; R[1] = -(b * i - c * h) * id; R[2] = (b * f - c * e) * id;
    R[3] = B * id; R[4] = (a * i - c * g) * id;  R[5] = -(a * f - c * d) * id;
    R[6] = C * id; R[7] = -(a * h - b * g) * id; R[8] = (a * e - b * d) * id;
    const float tx = v[12], ty = v[13], tz = v[14];
    o[0] = -(R[0] * tx + R[1] * ty + R[2] * tz);
    o[1] = -(R[3] * tx + R[4] * ty + R[5] * tz);
    o[2] = -(R[6] * tx + R[7] * ty + R[8] * tz);
}

__device__ __forceinline__ float finite_or_zero(float x) { return (x == x && fabsf(x) <= 3.402823466e38f) ? x : 0.f; }

__device__ __forceinline__ float surf_depth_at(const float* __restrict__ allmap, size_t HW, size_t pix, float ratio) {
    const float expected = finite_or_zero(allmap[pix] / allmap[HW + pix]);
    const float median = finite_or_zero(allmap[5 * HW + pix]);
    return expected * (1.f - ratio) + ratio * median;
}

__device__ __forceinline__ void point_at(const PostCam& cam, const float R[9], const float o[3], int x, int y, float depth, float p[3]) {
    const float dx = ((float)x - 0.5f * (float)cam.W) / cam.fx, dy = ((float)y - 0.5f * (float)cam.H) / cam.fy;
    p[0] = depth * (R[0] * dx + R[1] * dy + R[2]) + o[0];
    p[1] = depth * (R[3] * dx + R[4] * dy + R[5]) + o[1];
    p[2] = depth * (R[6] * dx + R[7] * dy + R[8]) + o[2];
}

__global__ __launch_bounds__(256) void postprocess_forward_kernel(PostCam cam, const float* __restrict__ allmap,
                                                                  float* __restrict__ rend_normal, float* __restrict__ surf_depth,
                                                                  float* __restrict__ surf_normal, float* __restrict__ surf_point) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cam.W || y >= cam.H) return;
    const size_t HW = (size_t)cam.W * cam.H, pix = (size_t)y * cam.W + x;
    float R[9], o[3];
    cam_to_world(cam.view, R, o);
    const float n0 = allmap[2 * HW + pix], n1 = allmap[3 * HW + pix], n2 = allmap[4 * HW + pix];
    rend_normal[pix] = R[0] * n0 + R[1] * n1 + R[2] * n2;
    rend_normal[HW + pix] = R[3] * n0 + R[4] * n1 + R[5] * n2;
    rend_normal[2 * HW + pix] = R[6] * n0 + R[7] * n1 + R[8] * n2;
    const float d = surf_depth_at(allmap, HW, pix, cam.depth_ratio);
    surf_depth[pix] = d;
    float p[3];
    point_at(cam, R, o, x, y, d, p);
    surf_point[pix] = p[0]; surf_point[HW + pix] = p[1]; surf_point[2 * HW + pix] = p[2];
    float sn[3] = {0.f, 0.f, 0.f};
    if (x > 0 && y > 0 && x < cam.W - 1 && y < cam.H - 1) {
        float pu[3], pd[3], pl[3], pr[3];
        point_at(cam, R, o, x, y + 1, surf_depth_at(allmap, HW, pix + cam.W, cam.depth_ratio), pd);
        point_at(cam, R, o, x, y - 1, surf_depth_at(allmap, HW, pix - cam.W, cam.depth_ratio), pu);
        point_at(cam, R, o, x + 1, y, surf_depth_at(allmap, HW, pix + 1, cam.depth_ratio), pr);
        point_at(cam, R, o, x - 1, y, surf_depth_at(allmap, HW, pix - 1, cam.depth_ratio), pl);
        const float ax = pd[0] - pu[0], ay = pd[1] - pu[1], az = pd[2] - pu[2];   // dx (rows)
        const float bx = pr[0] - pl[0], by = pr[1] - pl[1], bz = pr[2] - pl[2];   // dy (columns)
        const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const float inv = 1.f / fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
        const float alpha = allmap[HW + pix];
        sn[0] = cx * inv * alpha; sn[1] = cy * inv * alpha; sn[2] = cz * inv * alpha;
    }
    surf_normal[pix] = sn[0]; surf_normal[HW + pix] = sn[1]; surf_normal[2 * HW + pix] = sn[2];
}

// backward pass 1: per interior pixel, dL/d(dx), dL/d(dy) of its cross product -> G[6,H,W] (zeros on the border)
__global__ __launch_bounds__(256) void postprocess_backward_stencil_kernel(PostCam cam, const float* __restrict__ allmap,
                                                                           const float* __restrict__ g_surf_normal, float* __restrict__ G) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cam.W || y >= cam.H) return;
    const size_t HW = (size_t)cam.W * cam.H, pix = (size_t)y * cam.W + x;
    float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
    if (g_surf_normal && x > 0 && y > 0 && x < cam.W - 1 && y < cam.H - 1) {
        float R[9], o[3];
        cam_to_world(cam.view, R, o);
        float pu[3], pd[3], pl[3], pr[3];
        point_at(cam, R, o, x, y + 1, surf_depth_at(allmap, HW, pix + cam.W, cam.depth_ratio), pd);
        point_at(cam, R, o, x, y - 1, surf_depth_at(allmap, HW, pix - cam.W, cam.depth_ratio), pu);
        point_at(cam, R, o, x + 1, y, surf_depth_at(allmap, HW, pix + 1, cam.depth_ratio), pr);
        point_at(cam, R, o, x - 1, y, surf_depth_at(allmap, HW, pix - 1, cam.depth_ratio), pl);
        const float a[3] = {pd[0] - pu[0], pd[1] - pu[1], pd[2] - pu[2]};
        const float b[3] = {pr[0] - pl[0], pr[1] - pl[1], pr[2] - pl[2]};
        const float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const float norm = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const float alpha = allmap[HW + pix];
        const float gn[3] = {g_surf_normal[pix] * alpha, g_surf_normal[HW + pix] * alpha, g_surf_normal[2 * HW + pix] * alpha};
        float gc[3];
        if (norm > 1e-12f) {   // n = c / |c|
            const float inv = 1.f / norm;
            const float n[3] = {c[0] * inv, c[1] * inv, c[2] * inv};
            const float dot = n[0] * gn[0] + n[1] * gn[1] + n[2] * gn[2];
            gc[0] = (gn[0] - n[0] * dot) * inv; gc[1] = (gn[1] - n[1] * dot) * inv; gc[2] = (gn[2] - n[2] * dot) * inv;
        } else {               // n = c / eps
            gc[0] = gn[0] * 1e12f; gc[1] = gn[1] * 1e12f; gc[2] = gn[2] * 1e12f;
        }
        // c = a x b :  dL/da = b x gc ,  dL/db = gc x a
        ga[0] = b[1] * gc[2] - b[2] * gc[1]; ga[1] = b[2] * gc[0] - b[0] * gc[2]; ga[2] = b[0] * gc[1] - b[1] * gc[0];
        gb[0] = gc[1] * a[2] - gc[2] * a[1]; gb[1] = gc[2] * a[0] - gc[0] * a[2]; gb[2] = gc[0] * a[1] - gc[1] * a[0];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { G[k * HW + pix] = ga[k]; G[(3 + k) * HW + pix] = gb[k]; }
}

// backward pass 2: gather the stencil, chain to allmap channels 0, 1, 2-4, 5 (channel 6 gets 0)
__global__ __launch_bounds__(256) void postprocess_backward_gather_kernel(PostCam cam, const float* __restrict__ allmap,
                                                                          const float* __restrict__ g_rend_normal,
                                                                          const float* __restrict__ g_surf_depth,
                                                                          const float* __restrict__ g_surf_point, const float* __restrict__ G,
                                                                          float* __restrict__ g_allmap) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cam.W || y >= cam.H) return;
    const size_t HW = (size_t)cam.W * cam.H, pix = (size_t)y * cam.W + x;
    float R[9], o[3];
    cam_to_world(cam.view, R, o);
    float gp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = g_surf_point ? g_surf_point[k * HW + pix] : 0.f;
        if (y > 0) v += G[k * HW + pix - cam.W];            // this pixel is the "+row" neighbour of (y-1, x)
        if (y < cam.H - 1) v -= G[k * HW + pix + cam.W];    //                 "-row"               (y+1, x)
        if (x > 0) v += G[(3 + k) * HW + pix - 1];          //                 "+col"               (y, x-1)
        if (x < cam.W - 1) v -= G[(3 + k) * HW + pix + 1];  //                 "-col"               (y, x+1)
        gp[k] = v;
    }
    const float dx = ((float)x - 0.5f * (float)cam.W) / cam.fx, dy = ((float)y - 0.5f * (float)cam.H) / cam.fy;
    const float rd[3] = {R[0] * dx + R[1] * dy + R[2], R[3] * dx + R[4] * dy + R[5], R[6] * dx + R[7] * dy + R[8]};
    const float gd = (g_surf_depth ? g_surf_depth[pix] : 0.f) + gp[0] * rd[0] + gp[1] * rd[1] + gp[2] * rd[2];
    const float a0 = allmap[pix], alpha = allmap[HW + pix], med = allmap[5 * HW + pix];
    const float ratio = a0 / alpha;
    const bool e_ok = (ratio == ratio) && fabsf(ratio) <= 3.402823466e38f, m_ok = (med == med) && fabsf(med) <= 3.402823466e38f;
    const float ge = e_ok ? gd * (1.f - cam.depth_ratio) : 0.f;
    g_allmap[pix] = e_ok ? ge / alpha : 0.f;   // (torch yields NaN = 0 * inf at alpha == 0; empty pixels get a clean 0 here)
    g_allmap[HW + pix] = e_ok ? -ge * ratio / alpha : 0.f;
    g_allmap[5 * HW + pix] = m_ok ? gd * cam.depth_ratio : 0.f;
    g_allmap[6 * HW + pix] = 0.f;
    float gn[3] = {0.f, 0.f, 0.f};
    if (g_rend_normal) { gn[0] = g_rend_normal[pix]; gn[1] = g_rend_normal[HW + pix]; gn[2] = g_rend_normal[2 * HW + pix]; }
    g_allmap[2 * HW + pix] = R[0] * gn[0] + R[3] * gn[1] + R[6] * gn[2];   // R^T g
    g_allmap[3 * HW + pix] = R[1] * gn[0] + R[4] * gn[1] + R[7] * gn[2];
    g_allmap[4 * HW + pix] = R[2] * gn[0] + R[5] * gn[1] + R[8] * gn[2];
}

static dim3 post_grid(int W, int H) { return dim3((W + 63) / 64, (H + 3) / 4); }

hipError_t launch_postprocess_forward(const PostCam& cam, const float* allmap, float* rend_normal, float* surf_depth,
                                      float* surf_normal, float* surf_point, hipStream_t s) {
    hipLaunchKernelGGL(postprocess_forward_kernel, post_grid(cam.W, cam.H), dim3(256), 0, s, cam, allmap, rend_normal, surf_depth,
                       surf_normal, surf_point);
    return hipGetLastError();
}

hipError_t launch_postprocess_backward(const PostCam& cam, const float* allmap, const float* g_rend_normal, const float* g_surf_depth,
                                       const float* g_surf_normal, const float* g_surf_point, float* scratch6, float* g_allmap,
                                       hipStream_t s) {
    hipLaunchKernelGGL(postprocess_backward_stencil_kernel, post_grid(cam.W, cam.H), dim3(256), 0, s, cam, allmap, g_surf_normal, scratch6);
    hipLaunchKernelGGL(postprocess_backward_gather_kernel, post_grid(cam.W, cam.H), dim3(256), 0, s, cam, allmap, g_rend_normal,
                       g_surf_depth, g_surf_point, scratch6, g_allmap);
    return hipGetLastError();
}

}  // namespace sr
