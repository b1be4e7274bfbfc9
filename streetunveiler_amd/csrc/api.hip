// api.hip -- the C-ABI (include/surfel_raster.h): buffer layouts, argument checks, stage sequencing.
// No torch types, no exceptions across the boundary, nothing allocated persistently.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <atomic>
#include <mutex>

#include "common.h"

namespace sr {
// preprocess.hip
hipError_t launch_preprocess_forward(int P, const FrameDev& f, const SrGaussians& g, float4* recs, uint32_t* depth_keys,
                                     uint32_t* tiles_touched, uint2* rect, uint8_t* clamped, int32_t* radii, hipStream_t s);
hipError_t launch_preprocess_backward(int P, const FrameDev& f, const SrGaussians& g, const int32_t* radii,
                                      const uint8_t* clamped, const float4* recs, const float4* inst_grads, const uint8_t* written,
                                      const uint32_t* tiles_touched, const SrGradients& out, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
hipError_t launch_sh_gradient_expand(int P, int M, int deg, int V, const float* means3D, const float* campos, const float* gc,
                                     float* dL_dsh, hipStream_t s);
// binning.hip
size_t depth_sort_temp_bytes(int P);
size_t tile_scan_temp_bytes(int P);
size_t expand_x_hist_bytes(int P, int tiles_x);
size_t expand_y_hist_bytes(uint32_t D, int tiles_y);
hipError_t run_depth_sort(int P, const uint32_t* depth_keys, const uint2* rect, uint32_t* sorted_keys,
                          uint32_t* sorted_gid, uint2* rect_sorted, void* temp, size_t temp_bytes, int rank_mode, int tiles_x, int tiles_y,
                          const uint32_t* n_visible, hipStream_t s);
hipError_t run_tile_count_scan(int P, const uint32_t* tiles_touched, uint32_t* first, void* block_base, size_t base_bytes, uint32_t* total_host,
                               hipStream_t s);
hipError_t run_expand_columns(int P, int tiles_x, int n_tiles, const uint2* rect_sorted, const uint32_t* sorted_gid, uint2* columns,
                              uint32_t* n_columns, uint32_t* hist, uint32_t* row_total, uint32_t* tile_counts, int rank_mode, const uint32_t* n_visible,
                              hipStream_t s);
hipError_t run_expand_rows(uint32_t D, int tiles_x, int tiles_y, const uint2* columns, const uint32_t* n_columns, uint32_t* hist, uint32_t* row_total,
                           uint32_t* point_list, uint32_t* tile_counts, int rank_mode, hipStream_t s);
hipError_t run_tile_ranges_order(int n_tiles, const uint32_t* tile_counts, uint2* ranges, uint32_t* order, hipStream_t s);
hipError_t run_capacity_guard(uint32_t* counts, uint32_t capacity, hipStream_t s);
hipError_t launch_zero_bytes(void* p, size_t n, hipStream_t s);
// render.hip
hipError_t launch_render_forward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                 const float* extra, float* out_color, float* out_allmap, float* final_T, uint32_t* n_contrib, uint16_t* hit_mask, int flags,
                                 unsigned long long* counters, const uint32_t* frame_counts, hipStream_t s);
hipError_t launch_render_backward(const FrameDev& f, const uint2* ranges, const uint32_t* tile_order, const uint32_t* point_list, const float4* recs,
                                  const float* extra, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                  const float* dL_dallmap, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, bool precomp_color_grads, hipStream_t s, int coop_mode);
hipError_t launch_pair_decisions(const FrameDev& f, const uint2* ranges, const uint32_t* point_list, const float4* recs,
                                 unsigned long long* valid_bits, unsigned long long* use3d_bits, hipStream_t s);
hipError_t launch_class_partition(int P, int n_tiles, int n_classes, const float* class_cols, const int32_t* class_i32, const uint2* ranges,
                                  const uint32_t* point_list, uint8_t* ids, uint32_t* cls_list, uint2* cls_ranges, hipStream_t s);
hipError_t launch_class_forward(const FrameDev& f, int n_classes, const uint2* cls_ranges, const uint32_t* tile_order, const uint32_t* cls_list,
                                const float4* recs, float* out_dist, float* cls_state, uint32_t* cls_last, uint32_t* tile_total, uint16_t* hit_mask,
                                int cull, hipStream_t s);
hipError_t launch_class_backward(const FrameDev& f, int n_classes, const uint2* cls_ranges, const uint32_t* tile_order, const uint32_t* cls_list,
                                 const float4* recs, const float* cls_state, const uint32_t* cls_last, const uint32_t* tile_total,
                                 const float* dL_ddist, const uint16_t* hit_mask, float4* inst_grads, uint8_t* written, int shared_rec_quads, hipStream_t s);
hipError_t launch_color_gradients(int P, const FrameDev& f, const int32_t* radii, const uint8_t* clamped, const float4* recs, const float4* inst_grads,
                                  const uint8_t* written, const uint32_t* tiles_touched, bool mask_clamped, float* dL_dcolors, hipStream_t s);
// radix_sort.hip
size_t radix_sort_temp_bytes(uint32_t n);
hipError_t radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                            int total_bits, void* temp, size_t temp_bytes, hipStream_t s, const uint2* aux_src, uint2* aux_out, int rank_mode,
                            int rect_bx = 0, int rect_by = 0, const uint32_t* n_live = nullptr);
hipError_t launch_rank_selfcheck(uint32_t* result, hipStream_t s);
hipError_t lds_atomic_ranks(const uint32_t* digits, uint32_t* ranks, uint32_t n, int bins, hipStream_t s);
// knn.hip
size_t knn_workspace_bytes(int nq, int nr);
hipError_t knn_mean_dist2(int nq, const float* query, int nr, const float* reference, int K, int take_sqrt, float* out, void* ws,
                          size_t ws_bytes, int rank_mode, hipStream_t s);
// postprocess.hip
struct PostCam { int W, H; float fx, fy, depth_ratio; const float* view; };
hipError_t launch_postprocess_forward(const PostCam& cam, const float* allmap, float* rend_normal, float* surf_depth,
                                      float* surf_normal, float* surf_point, hipStream_t s);
hipError_t launch_postprocess_backward(const PostCam& cam, const float* allmap, const float* g_rend_normal, const float* g_surf_depth,
                                       const float* g_surf_normal, const float* g_surf_point, float* scratch6, float* g_allmap,
                                       hipStream_t s);
}  // namespace sr

using namespace sr;

namespace {

thread_local char g_err[512] = "";
// The only process-wide state of the library is this profiling aid (everything that changes what a call does is a field
// of that call's SrFrame): autograd runs the backward on its own thread, so the rings are shared and mutex-protected.
std::atomic<int> g_timing{0};
std::mutex g_ring_mu;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define SR_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return fail(SR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                               \
    } while (0)

// Stage timing: a ring of HIP event pairs per stage, recorded on the caller's stream (no sync while
// recording); sr_stage_stats() synchronises on the recorded events and returns total ms + launch count.
constexpr int kEvRing = 512;
struct EvRing {
    hipEvent_t ev[kEvRing][2];
    bool closed[kEvRing];   // the end event of the slot has been recorded (a reader skips a pair whose timer is still open)
    int created = 0, used = 0;
};
EvRing g_ring[SR_STAGE_COUNT];

struct StageTimer {
    int stage; hipStream_t s; int slot = -1;
    StageTimer(int st, hipStream_t stream) : stage(st), s(stream) {
        const int mode = g_timing.load();   // 0 off, 1 every stage, otherwise a bit mask of stages (bit 1 << stage, shifted by one)
        if (!mode || (mode != 1 && !((mode >> 1) & (1 << stage)))) return;
        std::lock_guard<std::mutex> lk(g_ring_mu);
        EvRing& r = g_ring[stage];
        if (r.used >= kEvRing) return;  // ring full: stop recording (stats stay valid for the recorded part)
        if (r.used >= r.created) {
            if (hipEventCreate(&r.ev[r.created][0]) != hipSuccess || hipEventCreate(&r.ev[r.created][1]) != hipSuccess) return;
            ++r.created;
        }
        slot = r.used++;   // reserved here: a timer started meanwhile on another thread (autograd's backward) gets the next one
        r.closed[slot] = false;
        (void)hipEventRecord(r.ev[slot][0], s);   // (two records per stage: each costs ~5 us of stream time)
    }
    ~StageTimer() {
        if (slot < 0) return;
        std::lock_guard<std::mutex> lk(g_ring_mu);
        if (hipEventRecord(g_ring[stage].ev[slot][1], s) == hipSuccess) g_ring[stage].closed[slot] = true;
    }
};

int debug_sync(const SrFrame* frame, hipStream_t s, const char* what) {
    if (!frame->debug) return SR_OK;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(SR_ERR_HIP, "[debug] after %s: %s", what, hipGetErrorString(e));
    return SR_OK;
}

// One 64-B pinned host block per calling thread (with one event per (thread, device) the only things this library keeps): word 0 =
// the DMA target of the num_rendered read-back, word 8 = the result word of the rank self-check.
// (portable + mapped: valid on every device of the process, whichever is current when the thread first calls in.  Never freed: 64 B per
// calling thread, and a destructor at thread / process exit could run after the HIP runtime has been torn down.)
uint32_t* pinned_words() {
    static thread_local uint32_t* pinned = nullptr;
    if (!pinned && hipHostMalloc(reinterpret_cast<void**>(&pinned), 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { pinned = nullptr; (void)hipGetLastError(); }
    return pinned;
}

// The device a stream belongs to (the cache below is per device: the CURRENT device may be another one); the null stream -> current device.
int stream_device(hipStream_t s, int* dev) {
    hipDevice_t d = 0;
    if (s && hipStreamGetDevice(s, &d) == hipSuccess) { *dev = (int)d; return SR_OK; }
    (void)hipGetLastError();
    SR_HIP(hipGetDevice(dev));
    return SR_OK;
}

// How the sort / partition kernels rank items inside a wave (common.h take_run_slot).  The fast path relies on the lane order of LDS
// atomic returns, which gfx950 delivers but no document promises -- so the first call on every device runs rank_selfcheck_kernel
// (~20 us) and the answer is cached per device for the life of the process: kRankAtomic, else the match-any ballots (kRankBallot),
// else nothing this library can sort with (SR_ERR_UNSUPPORTED).  SR_FLAG_BALLOT_RANKING forces the ballots for one call.
constexpr int kMaxDevices = 64;
std::atomic<int> g_rank_mode[kMaxDevices];
int rank_mode(hipStream_t s, bool force_ballot, int* mode) {
    int dev = 0;
    { const int rc = stream_device(s, &dev); if (rc != SR_OK) return rc; }
    if (dev < 0 || dev >= kMaxDevices) return fail(SR_ERR_UNSUPPORTED, "device index %d beyond %d", dev, kMaxDevices);
    int m = g_rank_mode[dev].load(std::memory_order_acquire);
    if (m == kRankUnknown) {
        // the self-check waits on the stream: not possible while the stream is being captured into a graph -- warm up first
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(SR_ERR_UNSUPPORTED, "the stream is capturing and device %d has not run the rank self-check yet: call sr_rank_mode(stream) once before the capture", dev);
        (void)hipGetLastError();
        struct Scratch { uint32_t* p = nullptr; ~Scratch() { if (p) (void)hipFree(p); } } scratch;   // freed on every path out of here
        uint32_t* host = pinned_words();
        uint32_t* result = nullptr;
        const bool temp = !(host && hipHostGetDevicePointer(reinterpret_cast<void**>(&result), host + 8, 0) == hipSuccess && result);
        if (!temp) host[8] = 0;
        else { (void)hipGetLastError(); SR_HIP(hipMalloc(reinterpret_cast<void**>(&scratch.p), 4)); result = scratch.p; SR_HIP(hipMemsetAsync(result, 0, 4, s)); }
        SR_HIP(launch_rank_selfcheck(result, s));
        uint32_t r = 0;
        if (temp) { SR_HIP(hipMemcpyAsync(&r, result, 4, hipMemcpyDeviceToHost, s)); SR_HIP(hipStreamSynchronize(s)); }
        else { SR_HIP(hipStreamSynchronize(s)); r = *reinterpret_cast<volatile uint32_t*>(host + 8); }
        if (!(r & 0x100u)) return fail(SR_ERR_HIP, "the rank self-check kernel did not report back");
        m = (r & 1u) ? kRankAtomic : ((r & 2u) ? kRankBallot : kRankNone);
        g_rank_mode[dev].store(m, std::memory_order_release);
    }
    if (m == kRankNone) return fail(SR_ERR_UNSUPPORTED, "neither LDS-atomic nor ballot ranking passes the self-check on device %d", dev);
    *mode = force_ballot ? (int)kRankBallot : m;
    return SR_OK;
}

// ---- buffer layouts ------------------------------------------------------------------------------
constexpr int kMaxTilesPerAxis = SR_MAX_TILES_PER_AXIS;   // binning.hip kXpMaxBins: 10-bit row / column fields, 1024-entry LDS histograms
struct GeomLayout {
    size_t recs, depth_keys, tiles_touched, rect, clamped, sorted_keys, sorted_gid, rect_sorted, first, sh_jac, block_base, base_bytes,
        n_scan_blocks, temp, temp_bytes, total;
};
GeomLayout geom_layout(int P) {
    GeomLayout L{};
    const size_t n = (size_t)(P > 0 ? P : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.recs = take(n * kRecFloats * 4);
    L.depth_keys = take(n * 4);
    L.tiles_touched = take(n * 4);
    L.rect = take(n * 8);
    L.clamped = take(n);
    L.sorted_keys = take(n * 4);
    L.sorted_gid = take(n * 4);
    L.rect_sorted = take(n * 8);
    L.first = take(n * 4);
    L.sh_jac = take(n * 36);
    L.base_bytes = tile_scan_temp_bytes(P);
    L.block_base = take(L.base_bytes);
    L.n_scan_blocks = (n + 2047) / 2048;   // the scan's block size (radix_sort.hip kRsTile)
    static thread_local int memo_P = -1;
    static thread_local size_t memo_bytes = 0;
    if (memo_P != P) {   // the depth sort's ping-pong + histogram; later pass X's [tile columns][blocks] histogram (any frame width)
        memo_bytes = depth_sort_temp_bytes(P);
        const size_t hx = align_up(expand_x_hist_bytes(P, kMaxTilesPerAxis), 256);
        if (hx > memo_bytes) memo_bytes = hx;
        memo_P = P;
    }
    L.temp_bytes = memo_bytes;
    L.temp = take(L.temp_bytes);
    L.total = off;
    return L;
}

struct BinLayout {
    size_t columns, point_list, hit_mask, ranges, order, tile_counts, row_total, n_columns, hist, hist_bytes, total;
};
BinLayout bin_layout(uint32_t D, int W, int H) {
    BinLayout L{};
    const size_t n = (size_t)(D > 0 ? D : 1);
    const int tiles = ((W + 7) / 8) * ((H + 7) / 8);   // sized for the smallest tile shape (8x8); the reference's 16x16 uses a quarter
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.columns = take(n * 8);   // column items of the expanding partition: at most D of them
    L.point_list = take(n * 4);
    L.hit_mask = take(n * 2);
    L.ranges = take((size_t)(tiles > 0 ? tiles : 1) * 8);
    L.order = take((size_t)(tiles > 0 ? tiles : 1) * 4);
    L.tile_counts = take((size_t)(tiles > 0 ? tiles : 1) * 4);
    L.row_total = take((size_t)kMaxTilesPerAxis * 4);
    L.n_columns = take(4);
    L.hist_bytes = expand_y_hist_bytes(D, (H + 7) / 8);
    L.hist = take(L.hist_bytes);
    L.total = off;
    return L;
}

struct ImgLayout { size_t final_T, n_contrib, total; };
ImgLayout img_layout(int W, int H) {
    ImgLayout L{};
    const size_t hw = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1);
    L.final_T = 0;
    L.n_contrib = align_up(hw * 3 * 4, 256);
    L.total = align_up(L.n_contrib + hw * 2 * 4, 256);
    return L;
}

int check_common(const SrFrame* frame, const SrGaussians* g) {
    if (!frame || !g) return fail(SR_ERR_INVALID_ARGUMENT, "frame / gaussians is NULL");
    if (frame->image_width <= 0 || frame->image_height <= 0) return fail(SR_ERR_INVALID_ARGUMENT, "bad image size %dx%d", frame->image_width, frame->image_height);
    if (g->P < 0) return fail(SR_ERR_INVALID_ARGUMENT, "P < 0");
    {
        const int tw = frame->tile_width > 0 ? frame->tile_width : kTile, th = frame->tile_height > 0 ? frame->tile_height : kTile;
        const bool known = (tw == 16 && th == 16) || (tw == 8 && th == 8) || (tw == 16 && th == 8) || (tw == 32 && th == 8) || (tw == 32 && th == 16);
        if (!known) return fail(SR_ERR_UNSUPPORTED, "tile shape %dx%d not in {8x8, 16x8, 16x16, 32x8, 32x16}", tw, th);
    }
    if (!frame->bg || !frame->viewmatrix || !frame->projmatrix || !frame->campos) return fail(SR_ERR_INVALID_ARGUMENT, "bg / viewmatrix / projmatrix / campos must be non-NULL device pointers");
    if (g->P > 0) {
        if (!g->means3D || !g->opacities) return fail(SR_ERR_INVALID_ARGUMENT, "means3D / opacities is NULL");
        if (g->color_channels == 9) {   // SH colour + six precomputed channels in one pass
            if (!g->shs || !g->colors_precomp) return fail(SR_ERR_INVALID_ARGUMENT, "9 colour channels need SHs AND a [P,6] precomputed colour array");
        } else if ((g->shs != nullptr) == (g->colors_precomp != nullptr)) return fail(SR_ERR_INVALID_ARGUMENT, "Please provide exactly one of either SHs or precomputed colors!");
        if (g->color_channels != 0 && g->color_channels != 3 && g->color_channels != 6 && g->color_channels != 9) return fail(SR_ERR_UNSUPPORTED, "color_channels %d not in {3, 6, 9}", g->color_channels);
        if (g->activations & ~(SR_ACT_EXP_SCALES | SR_ACT_SIGMOID_OPACITY | SR_ACT_NORMALIZE_ROTATIONS)) return fail(SR_ERR_UNSUPPORTED, "unknown activation bits 0x%x", g->activations);
        if (g->color_channels == 6 && g->shs) return fail(SR_ERR_INVALID_ARGUMENT, "6 colour channels need precomputed colors, not SHs");
        const bool sr_pair = g->scales != nullptr && g->rotations != nullptr;
        if ((g->scales != nullptr) != (g->rotations != nullptr) || sr_pair == (g->transMat_precomp != nullptr))
            return fail(SR_ERR_INVALID_ARGUMENT, "Please provide exactly one of either scale/rotation pair or precomputed transMat!");
        if (g->shs) {
            if (frame->sh_degree < 0 || frame->sh_degree > 3) return fail(SR_ERR_UNSUPPORTED, "sh_degree %d not in 0..3", frame->sh_degree);
            if (g->sh_coeffs < (frame->sh_degree + 1) * (frame->sh_degree + 1)) return fail(SR_ERR_INVALID_ARGUMENT, "shs has %d coefficients, degree %d needs %d", g->sh_coeffs, frame->sh_degree, (frame->sh_degree + 1) * (frame->sh_degree + 1));
        }
    }
    return SR_OK;
}

// The backward entry points read SrFrame.tanfovx / tanfovy (upstream's backward derives the image size from them: SR_BACKWARD_WH_FROM_FOCAL, the
// shipped default; the forward never did): a caller that left them 0 or NaN would get inf * 0 = NaN, (int)NaN -- undefined behaviour -- and
// a silently wrong viewport chain in K8.
int check_backward_frame(const SrFrame* frame) {
#if SR_BACKWARD_WH_FROM_FOCAL
    if (!(frame->tanfovx > 0.f) || !(frame->tanfovy > 0.f) || !std::isfinite(frame->tanfovx) || !std::isfinite(frame->tanfovy))
        return fail(SR_ERR_INVALID_ARGUMENT, "tanfovx / tanfovy must be finite and > 0 for a backward call (got %g, %g)", (double)frame->tanfovx, (double)frame->tanfovy);
#else
    (void)frame;
#endif
    return SR_OK;
}

FrameDev make_frame(const SrFrame* frame, const SrGaussians* g) {
    FrameDev f{};
    f.W = frame->image_width; f.H = frame->image_height;
    f.bw_W = f.W; f.bw_H = f.H;
#if SR_BACKWARD_WH_FROM_FOCAL
    if (frame->tanfovx > 0.f && frame->tanfovy > 0.f && std::isfinite(frame->tanfovx) && std::isfinite(frame->tanfovy)) {
        // upstream's backward: focal = size / (2 tanfov) (rasterizer_impl), then W = int(focal_x * tan_fovx * 2) in float32
        // (a forward call may leave the two fields unset -- it never reads bw_W / bw_H; the backward entry points insist: check_backward_frame)
        const float focal_x = (float)f.W / (2.0f * frame->tanfovx), focal_y = (float)f.H / (2.0f * frame->tanfovy);
        f.bw_W = (int)(focal_x * frame->tanfovx * 2); f.bw_H = (int)(focal_y * frame->tanfovy * 2);
    }
#endif
    f.tile_w = frame->tile_width > 0 ? frame->tile_width : kTile; f.tile_h = frame->tile_height > 0 ? frame->tile_height : kTile;
    f.inv_tile_w = 1.f / (float)f.tile_w; f.inv_tile_h = 1.f / (float)f.tile_h;
    f.tiles_x = (f.W + f.tile_w - 1) / f.tile_w; f.tiles_y = (f.H + f.tile_h - 1) / f.tile_h;
    f.sh_degree = frame->sh_degree; f.sh_coeffs = g->sh_coeffs;
    f.colors = g->color_channels == 6 ? 6 : (g->color_channels == 9 ? 9 : 3);
    f.activations = g->activations;
    f.scale_modifier = frame->scale_modifier;
    f.bg = frame->bg; f.view = frame->viewmatrix; f.proj = frame->projmatrix; f.campos = frame->campos;
    f.overflow = nullptr;   // (set by the backward entry points in capacity mode)
    return f;
}

template <class T> T* at(void* base, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }

}  // namespace

extern "C" {

int sr_abi_version(void) { return SR_ABI_VERSION; }
uint32_t sr_build_switches(void) { return SR_SWITCH_BITS; }
const char* sr_last_error(void) { return g_err; }

size_t sr_geom_bytes(int32_t P) { return geom_layout(P).total; }
size_t sr_binning_bytes(int32_t P, uint32_t num_rendered, int32_t W, int32_t H) { (void)P; return bin_layout(num_rendered, W, H).total; }
size_t sr_image_bytes(int32_t W, int32_t H) { return img_layout(W, H).total; }
static size_t record_bytes(int color_channels) { return (size_t)(color_channels == 9 ? kGradFloats + 4 : kGradFloats) * 4; }

size_t sr_backward_workspace_bytes(int32_t P, uint32_t num_rendered, int32_t color_channels) {
    (void)P;   // one gradient record (96 B; 112 B with 9 colour channels) + one "written" byte per (tile, Gaussian) duplicate
    const size_t n = (size_t)(num_rendered > 0 ? num_rendered : 1);
    return align_up(n * record_bytes(color_channels), 256) + align_up(n, 256);
}

int sr_geom_view(void* geom, size_t geom_bytes, int32_t P, SrGeomView* out) {
    if (!geom || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    const GeomLayout L = geom_layout(P);
    if (geom_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, L.total);
    out->splats = at<float>(geom, L.recs); out->depth_keys = at<uint32_t>(geom, L.depth_keys);
    out->tiles_touched = at<uint32_t>(geom, L.tiles_touched); out->clamped = at<uint8_t>(geom, L.clamped);
    out->sorted_gid = at<uint32_t>(geom, L.sorted_gid);
    out->frame_counts = at<uint32_t>(geom, L.block_base) + L.n_scan_blocks;
    return SR_OK;
}

int sr_binning_view(void* binning, size_t binning_bytes, int32_t P, uint32_t D, int32_t W, int32_t H, SrBinningView* out) {
    (void)P;
    if (!binning || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    const BinLayout L = bin_layout(D, W, H);
    if (binning_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "binning buffer %zu < %zu", binning_bytes, L.total);
    out->point_list = at<uint32_t>(binning, L.point_list);
    out->ranges = at<uint32_t>(binning, L.ranges); out->tile_order = at<uint32_t>(binning, L.order);
    return SR_OK;
}

int sr_image_view(void* image, size_t image_bytes, int32_t W, int32_t H, SrImageView* out) {
    if (!image || !out) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    const ImgLayout L = img_layout(W, H);
    if (image_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "image buffer %zu < %zu", image_bytes, L.total);
    out->final_T = at<float>(image, L.final_T); out->n_contrib = at<uint32_t>(image, L.n_contrib);
    return SR_OK;
}

int sr_forward_plan(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, int32_t* radii,
                    uint32_t* num_rendered_host, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    if (!num_rendered_host) return fail(SR_ERR_INVALID_ARGUMENT, "num_rendered_host is NULL");
    *num_rendered_host = 0;
    const int P = g->P;
    if (P == 0) return SR_OK;
    if (!geom || !radii) return fail(SR_ERR_INVALID_ARGUMENT, "geom / radii is NULL");
    const GeomLayout L = geom_layout(P);
    if (geom_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, L.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    FrameDev f = make_frame(frame, g);
    f.sh_jac = (frame->flags & SR_FLAG_FORWARD_ONLY) ? nullptr : at<float>(geom, L.sh_jac);   // (forward only: K8, its one reader, will not run)
    {
        StageTimer t(SR_STAGE_PREPROCESS, s);
        SR_HIP(launch_preprocess_forward(P, f, *g, at<float4>(geom, L.recs), at<uint32_t>(geom, L.depth_keys),
                                         at<uint32_t>(geom, L.tiles_touched), at<uint2>(geom, L.rect), at<uint8_t>(geom, L.clamped), radii, s));
    }
    if (int rc = debug_sync(frame, s, "preprocess_forward")) return rc;
    // D, the scan's grand total, is the one word the host reads back (the reference does the same between scan and duplicateWithKeys).
    // It travels through a pinned host word (one block per host thread; with one event per (thread, device) the only things this library
    // keeps): the last scan kernel stores it there itself when the word is mapped into the device's address space -- no copy kernel
    // between the scan and the host's wake-up -- else a DMA copy does.
    int sort_mode = kRankUnknown;
    if (int rc = rank_mode(s, (frame->flags & SR_FLAG_BALLOT_RANKING) != 0, &sort_mode)) return rc;   // (first call on a device: ~20 us self-check)
    const int depth_sort_mode = sort_mode | ((frame->flags & SR_FLAG_ONE_SWEEP_SORT) ? 0x100 : 0);   // (radix_sort.hip kSortOneSweepBit)
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) {
        // the sync-free forward: D stays on the device (SrGeomView.frame_counts); no pinned word, no event, no host wait -- nothing in this
        // call that a stream capture could not record
        {
            StageTimer t(SR_STAGE_SCAN, s);
            SR_HIP(run_tile_count_scan(P, at<uint32_t>(geom, L.tiles_touched), at<uint32_t>(geom, L.first), at<void>(geom, L.block_base), L.base_bytes, nullptr, s));
        }
        if (int rc = debug_sync(frame, s, "emission_scan")) return rc;
        {
            StageTimer t(SR_STAGE_DEPTH_SORT, s);
            SR_HIP(run_depth_sort(P, at<uint32_t>(geom, L.depth_keys), at<uint2>(geom, L.rect), at<uint32_t>(geom, L.sorted_keys),
                                  at<uint32_t>(geom, L.sorted_gid), at<uint2>(geom, L.rect_sorted), at<void>(geom, L.temp), L.temp_bytes, depth_sort_mode,
                                  f.tiles_x, f.tiles_y, at<uint32_t>(geom, L.block_base) + L.n_scan_blocks + 1, s));
        }
        *num_rendered_host = 0xFFFFFFFFu;   // unknown to the host
        return debug_sync(frame, s, "depth_sort");
    }
    uint32_t* pinned = pinned_words();
    uint32_t* pinned_dev = nullptr;
    if (pinned && (hipHostGetDevicePointer(reinterpret_cast<void**>(&pinned_dev), pinned, 0) != hipSuccess)) pinned_dev = nullptr;
    {
        StageTimer t(SR_STAGE_SCAN, s);
        SR_HIP(run_tile_count_scan(P, at<uint32_t>(geom, L.tiles_touched), at<uint32_t>(geom, L.first), at<void>(geom, L.block_base),
                                   L.base_bytes, pinned_dev, s));
    }
    if (int rc = debug_sync(frame, s, "emission_scan")) return rc;
    // The depth sort is queued BEHIND the read-back and the host waits for the read-back only, so the GPU sorts while the caller wakes
    // up, sizes the binning buffer from D and queues the second phase.
    static thread_local hipEvent_t copied_ev[kMaxDevices] = {};   // one marker per (calling thread, device): an event belongs to its device
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    hipEvent_t copied = nullptr;
    if (dev >= 0 && dev < kMaxDevices) {
        if (!copied_ev[dev] && hipEventCreateWithFlags(&copied_ev[dev], hipEventDisableTiming) != hipSuccess) copied_ev[dev] = nullptr;
        copied = copied_ev[dev];
    }
    if (!pinned_dev) {
        uint32_t* dst = pinned ? pinned : num_rendered_host;
        SR_HIP(hipMemcpyAsync(dst, at<uint32_t>(geom, L.block_base) + L.n_scan_blocks, 4, hipMemcpyDeviceToHost, s));   // the scan's grand total
    }
    if (copied && pinned && hipEventRecord(copied, s) != hipSuccess) copied = nullptr;   // (then: wait for the stream instead)
    {
        StageTimer t(SR_STAGE_DEPTH_SORT, s);
        SR_HIP(run_depth_sort(P, at<uint32_t>(geom, L.depth_keys), at<uint2>(geom, L.rect), at<uint32_t>(geom, L.sorted_keys),
                              at<uint32_t>(geom, L.sorted_gid), at<uint2>(geom, L.rect_sorted), at<void>(geom, L.temp), L.temp_bytes, depth_sort_mode,
                              f.tiles_x, f.tiles_y, at<uint32_t>(geom, L.block_base) + L.n_scan_blocks + 1, s));   // (+1: the scan's visible count)
    }
    if (int rc = debug_sync(frame, s, "depth_sort")) return rc;
    if (copied && pinned) SR_HIP(hipEventSynchronize(copied));
    else SR_HIP(hipStreamSynchronize(s));
    if (pinned) *num_rendered_host = *reinterpret_cast<volatile uint32_t*>(pinned);
    return SR_OK;
}

namespace {
// K3..K5 (duplicate emission, tile partition, tile ranges + dispatch order): shared by the blend forward and the per-class pass
int bin_duplicates(const SrFrame* frame, const SrGaussians* g, const FrameDev& f, void* geom, size_t geom_bytes, void* binning, const BinLayout& B,
                   uint32_t D, hipStream_t s, float4** recs_out) {
    const int P = g->P;
    const int n_tiles = f.tiles_x * f.tiles_y;
    *recs_out = nullptr;
    if (P > 0 && D > 0) {
        if (!geom) return fail(SR_ERR_INVALID_ARGUMENT, "geom is NULL");
        const GeomLayout L = geom_layout(P);
        if (geom_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, L.total);
        float4* recs = at<float4>(geom, L.recs);
        *recs_out = recs;
        if (f.tiles_x > kMaxTilesPerAxis || f.tiles_y > kMaxTilesPerAxis)
            return fail(SR_ERR_INVALID_ARGUMENT, "%d x %d tiles: at most %d per axis (use a larger tile)", f.tiles_x, f.tiles_y, kMaxTilesPerAxis);
        if (L.temp_bytes < expand_x_hist_bytes(P, f.tiles_x)) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom scratch too small for the column histogram");
        int sort_mode = kRankUnknown;
        if (int rc = rank_mode(s, (frame->flags & SR_FLAG_BALLOT_RANKING) != 0, &sort_mode)) return rc;
        if (frame->flags & SR_FLAG_BINNING_CAPACITY)   // D = the caller's capacity: does the frame fit?  (else: nothing is binned, the flag is set)
            SR_HIP(run_capacity_guard(at<uint32_t>(geom, L.block_base) + L.n_scan_blocks, D, s));
        {
            StageTimer t(SR_STAGE_EXPAND_X, s);
            SR_HIP(run_expand_columns(P, f.tiles_x, n_tiles, at<uint2>(geom, L.rect_sorted), at<uint32_t>(geom, L.sorted_gid), at<uint2>(binning, B.columns),
                                      at<uint32_t>(binning, B.n_columns), at<uint32_t>(geom, L.temp), at<uint32_t>(binning, B.row_total),
                                      at<uint32_t>(binning, B.tile_counts), sort_mode, at<uint32_t>(geom, L.block_base) + L.n_scan_blocks + 1, s));
        }
        if (int rc = debug_sync(frame, s, "expand_columns")) return rc;
        {
            StageTimer t(SR_STAGE_EXPAND_Y, s);
            SR_HIP(run_expand_rows(D, f.tiles_x, f.tiles_y, at<uint2>(binning, B.columns), at<uint32_t>(binning, B.n_columns),
                                   at<uint32_t>(binning, B.hist), at<uint32_t>(binning, B.row_total), at<uint32_t>(binning, B.point_list),
                                   at<uint32_t>(binning, B.tile_counts), sort_mode, s));
        }
        if (int rc = debug_sync(frame, s, "expand_rows")) return rc;
    } else {
        SR_HIP(launch_zero_bytes(at<uint32_t>(binning, B.tile_counts), sizeof(uint32_t) * (size_t)n_tiles, s));   // (no partition ran)
    }
    {
        StageTimer t(SR_STAGE_RANGES, s);
        SR_HIP(run_tile_ranges_order(n_tiles, at<uint32_t>(binning, B.tile_counts), at<uint2>(binning, B.ranges), at<uint32_t>(binning, B.order), s));
    }
    return debug_sync(frame, s, "tile_ranges");
}

// per-class state between the forward and the backward of the class pass.  (The class-ordered copy of the tile lists lives in the binning
// buffer's `columns` region -- the column items of the expanding partition are dead once pass Y has run -- and the class id bytes in the
// geometry buffer's `sh_jac` region: the class pass has no SH colour.)
struct ClassLayout { size_t state, last, tile_total, ranges, total; };
ClassLayout class_layout(int W, int H, int n_classes) {
    ClassLayout L{};
    const size_t hw = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1), n = (size_t)(n_classes > 0 ? n_classes : 1);
    const size_t tiles = (size_t)((W + 7) / 8) * (size_t)((H + 7) / 8);   // (sized for the smallest tile of the sweep, 8x8)
    L.state = 0;
    L.last = align_up(n * 3 * hw * 4, 256);
    L.tile_total = L.last + align_up(n * hw * 4, 256);
    L.ranges = L.tile_total + align_up((tiles > 0 ? tiles : 1) * n * 4, 256);
    L.total = L.ranges + align_up((tiles > 0 ? tiles : 1) * n * 8, 256);
    return L;
}

int check_class_pass(const SrFrame* frame, const SrGaussians* g, int n_classes) {
    if (int rc = check_common(frame, g)) return rc;
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_BINNING_CAPACITY serves the operator (sr_forward_* / sr_backward*), not the per-class pass");
    if (n_classes < 1 || n_classes > 6) return fail(SR_ERR_UNSUPPORTED, "n_classes %d not in 1..6", n_classes);
    const int tw = frame->tile_width > 0 ? frame->tile_width : kTile, th = frame->tile_height > 0 ? frame->tile_height : kTile;
    (void)tw; (void)th;   // (every tile shape check_common accepts: 8x8, 16x8, 16x16, 32x8, 32x16)
    if (g->P > 0 && (g->shs || !g->colors_precomp || (g->color_channels != 0 && g->color_channels != 3)))
        return fail(SR_ERR_INVALID_ARGUMENT, "the per-class distortion pass takes the class ids in colors_precomp[P,3] (column 0), no SHs");
    return SR_OK;
}
}  // namespace

int sr_forward_render(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning,
                      size_t binning_bytes, void* image, size_t image_bytes, uint32_t D, float* out_color,
                      float* out_allmap, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    const bool fwd_only = (frame->flags & SR_FLAG_FORWARD_ONLY) != 0;   // no backward follows: its state (image buffer, hit masks) is not written
    if (!binning || (!image && !fwd_only) || !out_color || !out_allmap) return fail(SR_ERR_INVALID_ARGUMENT, "binning / image / out_color / out_allmap is NULL");
    const int W = frame->image_width, H = frame->image_height;
    const BinLayout B = bin_layout(D, W, H);
    const ImgLayout I = img_layout(W, H);
    if (binning_bytes < B.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "binning buffer %zu < %zu", binning_bytes, B.total);
    if (!fwd_only && image_bytes < I.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "image buffer %zu < %zu", image_bytes, I.total);
    if (fwd_only && frame->blend_counters) return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_FORWARD_ONLY and blend_counters exclude each other");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const FrameDev f = make_frame(frame, g);
    if (frame->blend_counters && !(f.tile_w == 16 && f.tile_h == 16 && (f.colors == 3 || f.colors == 6)))
        return fail(SR_ERR_UNSUPPORTED, "blend_counters: the counting variant of the forward blend exists for the 16x16 tile with 3 or 6 colour channels only");
    float4* recs = nullptr;
    if (int rc = bin_duplicates(frame, g, f, geom, geom_bytes, binning, B, D, s, &recs)) return rc;
    {
        StageTimer t(SR_STAGE_BLEND_FWD, s);
        const bool rows = (frame->flags & SR_FLAG_ROW_MAPPED_FORWARD) != 0, quads = (frame->flags & SR_FLAG_QUADRANT_MAPPED_FORWARD) != 0;
        if (rows && quads) return fail(SR_ERR_INVALID_ARGUMENT, "SR_FLAG_ROW_MAPPED_FORWARD and SR_FLAG_QUADRANT_MAPPED_FORWARD exclude each other");
        if (rows && (!(f.tile_w == 16 && f.tile_h == 16 && f.colors == 3) || frame->blend_counters || (frame->flags & SR_FLAG_NO_QUADRANT_CULL)))
            return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_ROW_MAPPED_FORWARD: 16x16 tile, three colour channels, no counters, culling on");
        const bool cells = (frame->flags & SR_FLAG_ROW_BACKWARD) != 0;
        if (cells && (!(f.tile_w == 16 && f.tile_h == 16 && f.colors == 3) || frame->blend_counters || (frame->flags & SR_FLAG_NO_QUADRANT_CULL) || quads || fwd_only))
            return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_ROW_BACKWARD: 16x16 tile, three colour channels, no counters, culling on, the row-mapped forward, a backward to follow");
        const int flags = ((frame->flags & SR_FLAG_NO_QUADRANT_CULL) ? 0 : 1) | (frame->blend_counters ? 2 : 0) | (rows ? 4 : 0) | (quads ? 8 : 0) |
                          ((frame->flags & SR_FLAG_ONE_WAVE_BACKWARD) ? 16 : 0) | ((frame->flags & SR_FLAG_COOP_BACKWARD) ? 32 : 0) |   // (the few-tile kernels: never / always)
                          (cells ? 64 : 0);
        const GeomLayout GL = geom_layout(g->P);   // (D and the visible count, left in the geometry state by the emission scan)
        SR_HIP(launch_render_forward(f, at<uint2>(binning, B.ranges), at<uint32_t>(binning, B.order), at<uint32_t>(binning, B.point_list), recs, g->colors_precomp, out_color,
                                     out_allmap, fwd_only ? nullptr : at<float>(image, I.final_T), fwd_only ? nullptr : at<uint32_t>(image, I.n_contrib),
                                     fwd_only ? nullptr : at<uint16_t>(binning, B.hit_mask), flags,
                                     reinterpret_cast<unsigned long long*>(frame->blend_counters), at<uint32_t>(geom, GL.block_base) + GL.n_scan_blocks, s));
    }
    return debug_sync(frame, s, "render_forward");
}

size_t sr_class_image_bytes(int32_t W, int32_t H, int32_t n_classes) { return class_layout(W, H, n_classes).total; }

int sr_class_forward_render(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, void* geom, size_t geom_bytes, void* binning,
                            size_t binning_bytes, void* class_image, size_t class_image_bytes, uint32_t D, float* out_dist, void* stream) {
    if (int rc = check_class_pass(frame, g, n_classes)) return rc;
    if (!binning || !class_image || !out_dist) return fail(SR_ERR_INVALID_ARGUMENT, "binning / class_image / out_dist is NULL");
    if (g->P > 0 && !geom) return fail(SR_ERR_INVALID_ARGUMENT, "geom is NULL (the class ids of the P Gaussians are staged in it, even when no duplicate was emitted)");
    const int W = frame->image_width, H = frame->image_height;
    const BinLayout B = bin_layout(D, W, H);
    const ClassLayout C = class_layout(W, H, n_classes);
    if (binning_bytes < B.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "binning buffer %zu < %zu", binning_bytes, B.total);
    if (class_image_bytes < C.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "class image buffer %zu < %zu", class_image_bytes, C.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const FrameDev f = make_frame(frame, g);
    float4* recs = nullptr;
    if (int rc = bin_duplicates(frame, g, f, geom, geom_bytes, binning, B, D, s, &recs)) return rc;
    {
        // every tile list, stably partitioned by class: [class 0 by depth | class 1 by depth | ...] + a (begin, end) pair per (tile, class)
        StageTimer t(SR_STAGE_CLASS_PARTITION, s);
        const GeomLayout GL = geom_layout(g->P);
        SR_HIP(launch_class_partition(g->P, f.tiles_x * f.tiles_y, n_classes, g->colors_precomp, nullptr, at<uint2>(binning, B.ranges), at<uint32_t>(binning, B.point_list),
                                      g->P > 0 && geom ? at<uint8_t>(geom, GL.sh_jac) : nullptr, at<uint32_t>(binning, B.columns), at<uint2>(class_image, C.ranges), s));
    }
    if (int rc = debug_sync(frame, s, "class_partition")) return rc;
    {
        StageTimer t(SR_STAGE_CLASS_FWD, s);
        SR_HIP(launch_class_forward(f, n_classes, at<uint2>(class_image, C.ranges), at<uint32_t>(binning, B.order), at<uint32_t>(binning, B.columns), recs, out_dist,
                                    at<float>(class_image, C.state), at<uint32_t>(class_image, C.last), at<uint32_t>(class_image, C.tile_total),
                                    at<uint16_t>(binning, B.hit_mask), (frame->flags & SR_FLAG_NO_QUADRANT_CULL) ? 0 : 1, s));
    }
    return debug_sync(frame, s, "class_forward");
}

int sr_class_backward(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, const int32_t* radii, void* geom, size_t geom_bytes,
                      void* binning, size_t binning_bytes, void* class_image, size_t class_image_bytes, uint32_t D, const float* dL_ddist,
                      void* workspace, size_t workspace_bytes, const SrGradients* grads, void* stream) {
    if (int rc = check_class_pass(frame, g, n_classes)) return rc;
    if (int rc = check_backward_frame(frame)) return rc;
    if (!grads) return fail(SR_ERR_INVALID_ARGUMENT, "grads is NULL");
    const int P = g->P;
    if (P == 0) return SR_OK;
    if (!radii || !geom || !binning || !class_image || !dL_ddist || !workspace) return fail(SR_ERR_INVALID_ARGUMENT, "NULL buffer argument");
    const int W = frame->image_width, H = frame->image_height;
    const GeomLayout L = geom_layout(P);
    const BinLayout B = bin_layout(D, W, H);
    const ClassLayout C = class_layout(W, H, n_classes);
    if (geom_bytes < L.total || binning_bytes < B.total || class_image_bytes < C.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "state buffer too small");
    if (workspace_bytes < sr_backward_workspace_bytes(P, D, 3)) return fail(SR_ERR_BUFFER_TOO_SMALL, "workspace %zu < %zu", workspace_bytes, sr_backward_workspace_bytes(P, D, 3));
    hipStream_t s = static_cast<hipStream_t>(stream);
    FrameDev f = make_frame(frame, g);
    f.first = at<uint32_t>(geom, L.first); f.first_base = at<uint32_t>(geom, L.block_base); f.sh_jac = at<float>(geom, L.sh_jac);
    float4* inst_grads = static_cast<float4*>(workspace);
    uint8_t* written = static_cast<uint8_t*>(workspace) + align_up((size_t)(D > 0 ? D : 1) * record_bytes(3), 256);
    {
        StageTimer t(SR_STAGE_CLASS_BWD, s);
        if (D > 0) SR_HIP(launch_zero_bytes(written, D, s));
        if (D > 0)
            SR_HIP(launch_class_backward(f, n_classes, at<uint2>(class_image, C.ranges), at<uint32_t>(binning, B.order), at<uint32_t>(binning, B.columns), at<float4>(geom, L.recs),
                                         at<float>(class_image, C.state), at<uint32_t>(class_image, C.last), at<uint32_t>(class_image, C.tile_total), dL_ddist,
                                         at<uint16_t>(binning, B.hit_mask), inst_grads, written, 0, s));
    }
    if (int rc = debug_sync(frame, s, "class_backward")) return rc;
    {
        StageTimer t(SR_STAGE_PREPROCESS_BWD, s);
        SR_HIP(launch_preprocess_backward(P, f, *g, radii, at<uint8_t>(geom, L.clamped), at<float4>(geom, L.recs), inst_grads, written,
                                          at<uint32_t>(geom, L.tiles_touched), *grads, s));
    }
    return debug_sync(frame, s, "preprocess_backward");
}

// ---- the per-class pass on the binning of a colour pass (one plan, one binning, one K8 for both: SURVEY.md 8f N1 in full) -------------
namespace {
struct ClassSharedLayout { ClassLayout C; size_t ids, hit, total; };
ClassSharedLayout class_shared_layout(int P, int W, int H, int n_classes, uint32_t D) {
    ClassSharedLayout L{};
    L.C = class_layout(W, H, n_classes);
    L.ids = L.C.total;                                                   // class byte per Gaussian (the colour pass owns the sh_jac region here)
    L.hit = L.ids + align_up((size_t)(P > 0 ? P : 1), 256);              // this pass's own (entry, quadrant) hit masks: the colour pass keeps the binning buffer's
    L.total = L.hit + align_up((size_t)(D > 0 ? D : 1) * 2, 256);
    return L;
}
}  // namespace

size_t sr_class_shared_bytes(int32_t P, int32_t W, int32_t H, int32_t n_classes, uint32_t D) { return class_shared_layout(P, W, H, n_classes, D).total; }

int sr_class_forward_shared(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, const int32_t* classes, void* geom, size_t geom_bytes,
                            void* binning, size_t binning_bytes, void* class_state, size_t class_state_bytes, uint32_t D, float* out_dist, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_BINNING_CAPACITY serves the operator (sr_forward_* / sr_backward*), not the per-class pass");
    if (n_classes < 1 || n_classes > 6) return fail(SR_ERR_UNSUPPORTED, "n_classes %d not in 1..6", n_classes);
    if (!binning || !class_state || !out_dist || (g->P > 0 && !classes)) return fail(SR_ERR_INVALID_ARGUMENT, "binning / class_state / out_dist / classes is NULL");
    if (g->transMat_precomp) return fail(SR_ERR_UNSUPPORTED, "the per-class pass takes scales and rotations, not a precomputed transMat");
    const int W = frame->image_width, H = frame->image_height, P = g->P;
    const BinLayout B = bin_layout(D, W, H);
    const ClassSharedLayout S = class_shared_layout(P, W, H, n_classes, D);
    if (binning_bytes < B.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "binning buffer %zu < %zu", binning_bytes, B.total);
    if (class_state_bytes < S.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "class state buffer %zu < %zu", class_state_bytes, S.total);
    float4* recs = nullptr;
    if (P > 0 && D > 0) {
        if (!geom) return fail(SR_ERR_INVALID_ARGUMENT, "geom is NULL");
        const GeomLayout L = geom_layout(P);
        if (geom_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, L.total);
        recs = at<float4>(geom, L.recs);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const FrameDev f = make_frame(frame, g);
    {
        StageTimer t(SR_STAGE_CLASS_PARTITION, s);
        SR_HIP(launch_class_partition(P, f.tiles_x * f.tiles_y, n_classes, nullptr, classes, at<uint2>(binning, B.ranges), at<uint32_t>(binning, B.point_list),
                                      at<uint8_t>(class_state, S.ids), at<uint32_t>(binning, B.columns), at<uint2>(class_state, S.C.ranges), s));
    }
    if (int rc = debug_sync(frame, s, "class_partition")) return rc;
    {
        StageTimer t(SR_STAGE_CLASS_FWD, s);
        SR_HIP(launch_class_forward(f, n_classes, at<uint2>(class_state, S.C.ranges), at<uint32_t>(binning, B.order), at<uint32_t>(binning, B.columns), recs, out_dist,
                                    at<float>(class_state, S.C.state), at<uint32_t>(class_state, S.C.last), at<uint32_t>(class_state, S.C.tile_total),
                                    at<uint16_t>(class_state, S.hit), (frame->flags & SR_FLAG_NO_QUADRANT_CULL) ? 0 : 1, s));
    }
    return debug_sync(frame, s, "class_forward");
}

int sr_class_backward_shared(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, void* geom, size_t geom_bytes, void* binning,
                             size_t binning_bytes, void* class_state, size_t class_state_bytes, uint32_t D, const float* dL_ddist, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) return fail(SR_ERR_UNSUPPORTED, "SR_FLAG_BINNING_CAPACITY serves the operator (sr_forward_* / sr_backward*), not the per-class pass");
    if (int rc = check_backward_frame(frame)) return rc;
    if (n_classes < 1 || n_classes > 6) return fail(SR_ERR_UNSUPPORTED, "n_classes %d not in 1..6", n_classes);
    const int P = g->P;
    if (P == 0 || D == 0) return SR_OK;
    if (!geom || !binning || !class_state || !dL_ddist || !workspace) return fail(SR_ERR_INVALID_ARGUMENT, "NULL buffer argument");
    const int W = frame->image_width, H = frame->image_height;
    const GeomLayout L = geom_layout(P);
    const BinLayout B = bin_layout(D, W, H);
    const ClassSharedLayout S = class_shared_layout(P, W, H, n_classes, D);
    if (geom_bytes < L.total || binning_bytes < B.total || class_state_bytes < S.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "state buffer too small");
    if (workspace_bytes < sr_backward_workspace_bytes(P, D, g->color_channels)) return fail(SR_ERR_BUFFER_TOO_SMALL, "workspace %zu < %zu", workspace_bytes, sr_backward_workspace_bytes(P, D, g->color_channels));
    hipStream_t s = static_cast<hipStream_t>(stream);
    FrameDev f = make_frame(frame, g);
    f.first = at<uint32_t>(geom, L.first); f.first_base = at<uint32_t>(geom, L.block_base);
    // the records and flags sr_backward_blend of the SAME frame left in the workspace: this pass adds to them
    float4* inst_grads = static_cast<float4*>(workspace);
    uint8_t* written = static_cast<uint8_t*>(workspace) + align_up((size_t)D * record_bytes(g->color_channels), 256);
    {
        StageTimer t(SR_STAGE_CLASS_BWD, s);
        SR_HIP(launch_class_backward(f, n_classes, at<uint2>(class_state, S.C.ranges), at<uint32_t>(binning, B.order), at<uint32_t>(binning, B.columns), at<float4>(geom, L.recs),
                                     at<float>(class_state, S.C.state), at<uint32_t>(class_state, S.C.last), at<uint32_t>(class_state, S.C.tile_total), dL_ddist,
                                     at<uint16_t>(class_state, S.hit), inst_grads, written, (int)(record_bytes(g->color_channels) / 16), s));
    }
    return debug_sync(frame, s, "class_backward_shared");
}

namespace {
struct BackwardCtx {
    int P; GeomLayout L; BinLayout B; ImgLayout I; FrameDev f; float4* inst_grads; uint8_t* written; hipStream_t s;
};
// argument checks and buffer carving shared by the backward entry points
int backward_ctx(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                 void* image, size_t image_bytes, uint32_t D, void* workspace, size_t workspace_bytes, void* stream, BackwardCtx* c) {
    if (int rc = check_common(frame, g)) return rc;
    if (int rc = check_backward_frame(frame)) return rc;
    c->P = g->P;
    if (c->P == 0) return SR_OK;
    if (!geom || !binning || !image || !workspace) return fail(SR_ERR_INVALID_ARGUMENT, "NULL buffer argument");
    const int W = frame->image_width, H = frame->image_height;
    c->L = geom_layout(c->P); c->B = bin_layout(D, W, H); c->I = img_layout(W, H);
    if (geom_bytes < c->L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, c->L.total);
    if (binning_bytes < c->B.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "binning buffer %zu < %zu", binning_bytes, c->B.total);
    if (image_bytes < c->I.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "image buffer %zu < %zu", image_bytes, c->I.total);
    if (workspace_bytes < sr_backward_workspace_bytes(c->P, D, g->color_channels)) return fail(SR_ERR_BUFFER_TOO_SMALL, "workspace %zu < %zu", workspace_bytes, sr_backward_workspace_bytes(c->P, D, g->color_channels));
    c->s = static_cast<hipStream_t>(stream);
    c->f = make_frame(frame, g);
    c->f.first = at<uint32_t>(geom, c->L.first); c->f.first_base = at<uint32_t>(geom, c->L.block_base); c->f.sh_jac = at<float>(geom, c->L.sh_jac);
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) c->f.overflow = at<uint32_t>(geom, c->L.block_base) + c->L.n_scan_blocks + 2;   // (frame_counts[2])
    // per-(tile, Gaussian) gradient records in emission order (a Gaussian's duplicates are contiguous).  K7 writes a record --
    // and sets the slot's byte in `written` -- only where some pixel contributed; K8 looks at the byte before it touches the
    // record, so neither the records nor anything but these D bytes need clearing.
    c->inst_grads = static_cast<float4*>(workspace);
    c->written = static_cast<uint8_t*>(workspace) + align_up((size_t)(D > 0 ? D : 1) * record_bytes(g->color_channels), 256);
    return SR_OK;
}
}  // namespace

int sr_backward_blend(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                      void* image, size_t image_bytes, uint32_t D, const float* dL_dcolor, const float* dL_dallmap, void* workspace,
                      size_t workspace_bytes, void* stream) {
    BackwardCtx c;
    if (int rc = backward_ctx(frame, g, geom, geom_bytes, binning, binning_bytes, image, image_bytes, D, workspace, workspace_bytes, stream, &c)) return rc;
    if (c.P == 0) return SR_OK;
    if (!dL_dcolor || !dL_dallmap) return fail(SR_ERR_INVALID_ARGUMENT, "dL_dcolor / dL_dallmap is NULL");
    {
        StageTimer t(SR_STAGE_BLEND_BWD, c.s);
        if (D > 0) SR_HIP(launch_zero_bytes(c.written, D, c.s));
        if (D > 0)
            SR_HIP(launch_render_backward(c.f, at<uint2>(binning, c.B.ranges), at<uint32_t>(binning, c.B.order), at<uint32_t>(binning, c.B.point_list), at<float4>(geom, c.L.recs), g->colors_precomp,
                                          at<float>(image, c.I.final_T), at<uint32_t>(image, c.I.n_contrib), dL_dcolor, dL_dallmap, at<uint16_t>(binning, c.B.hit_mask), c.inst_grads, c.written,
                                          !(frame->flags & SR_FLAG_NO_PRECOMP_COLOR_GRAD), c.s,
                                          (frame->flags & SR_FLAG_ROW_BACKWARD) ? 3
                                          : ((frame->flags & SR_FLAG_ONE_WAVE_BACKWARD) ? 1 : ((frame->flags & SR_FLAG_COOP_BACKWARD) ? 2 : 0))));
    }
    return debug_sync(frame, c.s, "render_backward");
}

int sr_backward_colors(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes, uint32_t D,
                       void* workspace, size_t workspace_bytes, float* dL_dcolors, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    if (int rc = check_backward_frame(frame)) return rc;
    const int P = g->P;
    if (P == 0) return SR_OK;
    if (g->color_channels == 6 || g->color_channels == 9) return fail(SR_ERR_UNSUPPORTED, "sr_backward_colors serves the 3-channel pass");
    if (!radii || !geom || !workspace || !dL_dcolors) return fail(SR_ERR_INVALID_ARGUMENT, "NULL buffer argument");
    const GeomLayout L = geom_layout(P);
    if (geom_bytes < L.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "geom buffer %zu < %zu", geom_bytes, L.total);
    if (workspace_bytes < sr_backward_workspace_bytes(P, D, g->color_channels)) return fail(SR_ERR_BUFFER_TOO_SMALL, "workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FrameDev f = make_frame(frame, g);
    f.first = at<uint32_t>(geom, L.first); f.first_base = at<uint32_t>(geom, L.block_base); f.sh_jac = at<float>(geom, L.sh_jac);
    if (frame->flags & SR_FLAG_BINNING_CAPACITY) f.overflow = at<uint32_t>(geom, L.block_base) + L.n_scan_blocks + 2;
    const uint8_t* written = static_cast<uint8_t*>(workspace) + align_up((size_t)(D > 0 ? D : 1) * record_bytes(g->color_channels), 256);
    StageTimer t(SR_STAGE_PREPROCESS_BWD, s);
    SR_HIP(launch_color_gradients(P, f, radii, at<uint8_t>(geom, L.clamped), at<float4>(geom, L.recs), static_cast<const float4*>(workspace), written,
                                  at<uint32_t>(geom, L.tiles_touched), g->shs != nullptr, dL_dcolors, s));
    return debug_sync(frame, s, "color_gradients");
}

int sr_backward_geometry(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes,
                         void* binning, size_t binning_bytes, void* image, size_t image_bytes, uint32_t D, void* workspace,
                         size_t workspace_bytes, const SrGradients* grads, void* stream) {
    if (!grads) return fail(SR_ERR_INVALID_ARGUMENT, "grads is NULL");
    BackwardCtx c;
    if (int rc = backward_ctx(frame, g, geom, geom_bytes, binning, binning_bytes, image, image_bytes, D, workspace, workspace_bytes, stream, &c)) return rc;
    if (c.P == 0) return SR_OK;
    if (!radii) return fail(SR_ERR_INVALID_ARGUMENT, "radii is NULL");
    {
        StageTimer t(SR_STAGE_PREPROCESS_BWD, c.s);
        SR_HIP(launch_preprocess_backward(c.P, c.f, *g, radii, at<uint8_t>(geom, c.L.clamped), at<float4>(geom, c.L.recs), c.inst_grads, c.written,
                                          at<uint32_t>(geom, c.L.tiles_touched), *grads, c.s));
    }
    return debug_sync(frame, c.s, "preprocess_backward");
}

int sr_backward(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes,
                void* binning, size_t binning_bytes, void* image, size_t image_bytes, uint32_t D, const float* dL_dcolor,
                const float* dL_dallmap, void* workspace, size_t workspace_bytes, const SrGradients* grads, void* stream) {
    if (!grads) return fail(SR_ERR_INVALID_ARGUMENT, "grads is NULL");
    if (int rc = sr_backward_blend(frame, g, geom, geom_bytes, binning, binning_bytes, image, image_bytes, D, dL_dcolor, dL_dallmap, workspace, workspace_bytes, stream)) return rc;
    return sr_backward_geometry(frame, g, radii, geom, geom_bytes, binning, binning_bytes, image, image_bytes, D, workspace, workspace_bytes, grads, stream);
}

int sr_debug_pair_decisions(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                            uint32_t D, uint64_t* valid_bits, uint64_t* use3d_bits, void* stream) {
    if (int rc = check_common(frame, g)) return rc;
    if (g->P == 0 || D == 0) return SR_OK;
    if (!geom || !binning || !valid_bits || !use3d_bits) return fail(SR_ERR_INVALID_ARGUMENT, "NULL buffer argument");
    const GeomLayout L = geom_layout(g->P);
    const BinLayout B = bin_layout(D, frame->image_width, frame->image_height);
    if (geom_bytes < L.total || binning_bytes < B.total) return fail(SR_ERR_BUFFER_TOO_SMALL, "state buffer too small");
    const FrameDev f = make_frame(frame, g);
    SR_HIP(launch_pair_decisions(f, at<uint2>(binning, B.ranges), at<uint32_t>(binning, B.point_list), at<float4>(geom, L.recs),
                                 reinterpret_cast<unsigned long long*>(valid_bits), reinterpret_cast<unsigned long long*>(use3d_bits), static_cast<hipStream_t>(stream)));
    return SR_OK;
}

int sr_sh_gradient_expand(int32_t P, int32_t sh_coeffs, int32_t sh_degree, int32_t n_views, const float* means3D,
                          const float* campos, const float* dL_dcolors, float* dL_dsh, void* stream) {
    if (P < 0 || n_views < 1) return fail(SR_ERR_INVALID_ARGUMENT, "P < 0 or n_views < 1");
    if (sh_degree < 0 || sh_degree > 3) return fail(SR_ERR_UNSUPPORTED, "sh_degree %d not in 0..3", sh_degree);
    if (sh_coeffs < (sh_degree + 1) * (sh_degree + 1)) return fail(SR_ERR_INVALID_ARGUMENT, "shs has %d coefficients, degree %d needs %d", sh_coeffs, sh_degree, (sh_degree + 1) * (sh_degree + 1));
    if (P == 0) return SR_OK;
    if (!means3D || !campos || !dL_dcolors || !dL_dsh) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    SR_HIP(launch_sh_gradient_expand(P, sh_coeffs, sh_degree, n_views, means3D, campos, dL_dcolors, dL_dsh, static_cast<hipStream_t>(stream)));
    return SR_OK;
}

size_t sr_knn_workspace_bytes(int32_t n_query, int32_t n_reference) {
    return knn_workspace_bytes(n_query > 0 ? n_query : 0, n_reference > 0 ? n_reference : 0);
}

int sr_knn_mean_dist2(int32_t n_query, const float* query, int32_t n_reference, const float* reference, int32_t K,
                      int32_t take_sqrt, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (K != 3 && K != 10) return fail(SR_ERR_UNSUPPORTED, "K = %d, supported: 3 and 10", K);
    if (n_reference < 0 || (query && n_query < 0)) return fail(SR_ERR_INVALID_ARGUMENT, "negative point count");
    const int nq = query ? n_query : 0;
    if ((query ? nq : n_reference) == 0) return SR_OK;
    if (n_reference == 0) return fail(SR_ERR_INVALID_ARGUMENT, "empty reference cloud");
    if (!reference || !out || !workspace) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (workspace_bytes < knn_workspace_bytes(nq, n_reference)) return fail(SR_ERR_BUFFER_TOO_SMALL, "workspace %zu < %zu", workspace_bytes, knn_workspace_bytes(nq, n_reference));
    int sort_mode = kRankUnknown;
    if (int rc = rank_mode(static_cast<hipStream_t>(stream), false, &sort_mode)) return rc;
    SR_HIP(knn_mean_dist2(nq, query, n_reference, reference, K, take_sqrt, out, workspace, workspace_bytes, sort_mode, static_cast<hipStream_t>(stream)));
    return SR_OK;
}

int sr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                    void* stream) {
    (void)projmatrix;
    if (P < 0) return fail(SR_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return SR_OK;
    if (!means3D || !viewmatrix || !present) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    SR_HIP(launch_mark_visible(P, means3D, viewmatrix, present, static_cast<hipStream_t>(stream)));
    return SR_OK;
}

static int post_cam(int32_t W, int32_t H, float fovx, float fovy, float depth_ratio, const float* viewmatrix, PostCam* cam) {
    if (W <= 0 || H <= 0 || !viewmatrix) return fail(SR_ERR_INVALID_ARGUMENT, "bad image size / NULL viewmatrix");
    cam->W = W; cam->H = H;
    cam->fx = (float)W / (2.f * tanf(fovx * 0.5f)); cam->fy = (float)H / (2.f * tanf(fovy * 0.5f));
    cam->depth_ratio = depth_ratio; cam->view = viewmatrix;
    return SR_OK;
}

int sr_postprocess_forward(int32_t W, int32_t H, float fovx, float fovy, float depth_ratio, const float* viewmatrix,
                           const float* allmap, float* rend_normal, float* surf_depth, float* surf_normal, float* surf_point,
                           void* stream) {
    PostCam cam;
    if (int rc = post_cam(W, H, fovx, fovy, depth_ratio, viewmatrix, &cam)) return rc;
    if (!allmap || !rend_normal || !surf_depth || !surf_normal || !surf_point) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    SR_HIP(launch_postprocess_forward(cam, allmap, rend_normal, surf_depth, surf_normal, surf_point, static_cast<hipStream_t>(stream)));
    return SR_OK;
}

int sr_postprocess_backward(int32_t W, int32_t H, float fovx, float fovy, float depth_ratio, const float* viewmatrix,
                            const float* allmap, const float* g_rend_normal, const float* g_surf_depth, const float* g_surf_normal,
                            const float* g_surf_point, float* scratch6, float* g_allmap, void* stream) {
    PostCam cam;
    if (int rc = post_cam(W, H, fovx, fovy, depth_ratio, viewmatrix, &cam)) return rc;
    if (!allmap || !scratch6 || !g_allmap) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    SR_HIP(launch_postprocess_backward(cam, allmap, g_rend_normal, g_surf_depth, g_surf_normal, g_surf_point, scratch6, g_allmap,
                                       static_cast<hipStream_t>(stream)));
    return SR_OK;
}

int sr_debug_lds_atomic_ranks(const uint32_t* digits, uint32_t* ranks, uint32_t n, int bins, void* stream) {
    if (n > 0 && (!digits || !ranks)) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bins < 1 || bins > 1024 || (n % 256u) != 0u) return fail(SR_ERR_INVALID_ARGUMENT, "bins %d not in 1..1024 or n %u not a multiple of 256", bins, n);
    SR_HIP(lds_atomic_ranks(digits, ranks, n, bins, static_cast<hipStream_t>(stream)));
    return SR_OK;
}

int sr_rank_mode(void* stream) {
    int mode = kRankUnknown;
    if (int rc = rank_mode(static_cast<hipStream_t>(stream), false, &mode)) return rc;
    return mode;
}

size_t sr_debug_radix_sort_temp_bytes(uint32_t n) { return radix_sort_temp_bytes(n); }

int sr_debug_radix_sort(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                        int total_bits, void* temp, size_t temp_bytes, uint32_t flags, void* stream) {
    if (n > 0 && (!keys_in || !keys_out || !vals_out || !temp)) return fail(SR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (total_bits < 1 || total_bits > 32) return fail(SR_ERR_INVALID_ARGUMENT, "total_bits %d not in 1..32", total_bits);
    if (temp_bytes < radix_sort_temp_bytes(n)) return fail(SR_ERR_BUFFER_TOO_SMALL, "temp %zu < %zu", temp_bytes, radix_sort_temp_bytes(n));
    int sort_mode = kRankUnknown;
    if (int rc = rank_mode(static_cast<hipStream_t>(stream), (flags & SR_FLAG_BALLOT_RANKING) != 0, &sort_mode)) return rc;
    if (flags & SR_FLAG_ONE_SWEEP_SORT) sort_mode |= 0x100;   // (radix_sort.hip kSortOneSweepBit)
    SR_HIP(radix_sort_pairs(keys_in, vals_in, keys_out, vals_out, n, total_bits, temp, temp_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr, sort_mode));
    return SR_OK;
}

void sr_set_stage_timing(int enable) {
    std::lock_guard<std::mutex> lk(g_ring_mu);
    g_timing.store(enable);
    if (enable) for (auto& r : g_ring) r.used = 0;
}

int sr_stage_stats(int stage, float* total_ms, int* launches) {
    if (stage < 0 || stage >= SR_STAGE_COUNT || !total_ms || !launches) return fail(SR_ERR_INVALID_ARGUMENT, "bad stage / NULL output");
    std::lock_guard<std::mutex> lk(g_ring_mu);
    EvRing& r = g_ring[stage];
    float sum = 0.f;
    int done = 0;
    for (int i = 0; i < r.used; ++i) {
        if (!r.closed[i]) continue;
        ++done;
        float ms = 0.f;
        SR_HIP(hipEventSynchronize(r.ev[i][1]));
        SR_HIP(hipEventElapsedTime(&ms, r.ev[i][0], r.ev[i][1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = done;
    return SR_OK;
}

}  // extern "C"
