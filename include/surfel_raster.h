/*
 * surfel_raster.h -- C-ABI of the MI355X-native 2D-Gaussian (surfel) splatting rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of StreetUnveiler:
 *     gaussian_renderer.render -> diff_surfel_rasterization.GaussianRasterizer
 *         -> _C.rasterize_gaussians / _C.rasterize_gaussians_backward / _C.mark_visible
 * The reference binds that path through a torch C++ extension (`diff_surfel_rasterization._C`,
 * an un-vendored submodule: /root/reference/.gitmodules:9-12; only importer:
 * /root/reference/gaussian_renderer/__init__.py:11).  The entry points below are what that
 * extension's three functions bind to in this build; the python shim that reproduces the `_C`
 * call shape on top of them is diff_surfel_rasterization/__init__.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every `const float*` / buffer argument is a DEVICE pointer owned by the caller (a torch
 *     tensor kept alive by the autograd ctx) unless the name ends in `_host`.
 *   - the library allocates nothing persistent; all scratch lives in the three caller-owned state
 *     buffers (geom / binning / image) whose sizes come from the sr_*_bytes() queries -- the
 *     counterpart of the reference's resize-callback buffers geomBuffer / binningBuffer / imgBuffer
 *     (call shape: SURVEY.md 8b "Native signatures").
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - return value: 0 on success, negative SrStatus on failure; sr_last_error() gives the text of
 *     the last failure on the calling thread.  No exceptions cross the ABI.
 *   - tensor layouts are the operator's: float32, contiguous; means3D[P,3], opacities[P,1],
 *     scales[P,2], rotations[P,4] (r,x,y,z), shs[P,M,3] (coefficient-major), colors_precomp[P,NC],
 *     transMat_precomp[P,9]; images are planar [C,H,W].  NC = SrGaussians.color_channels: 3 as in the reference, 9 (see the
 *     struct), or 6
 *     (precomputed colours only) = two 3-channel passes over the same geometry folded into one -- what the reference's
 *     render_semantic does with two rasterizer calls (/root/reference/gaussian_renderer/__init__.py:386-431)
 *     (/root/reference/gaussian_renderer/__init__.py:56-138; allmap channel order :149-165).
 */
#ifndef SURFEL_RASTER_H
#define SURFEL_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_ABI_VERSION 10
#define SR_TILE 16            /* default 16x16 pixel tiles (upstream BLOCK_X/BLOCK_Y); see SrFrame.tile_width */
#define SR_SPLAT_FLOATS 20    /* floats per packed splat record (80 B) */
#define SR_GRAD_FLOATS 24     /* floats per gradient record (96 B) */
#define SR_MAX_TILES_PER_AXIS 1024 /* tiles per image axis the binning handles (10-bit row / column fields of the expanding partition):
                                    * frames up to 16384 px per axis with the 16x16 tile, 8192 px with 8x8, 32768 px wide with 32x16;
                                    * larger frames are refused with SR_ERR_INVALID_ARGUMENT (pick a larger tile) */

/* SrGaussians.activations == the GaussianModel activations
 * (/root/reference/scene/gaussian_model.py:63-75, getters :101-123) */
#define SR_ACT_EXP_SCALES 1          /* scales = exp(raw) */
#define SR_ACT_SIGMOID_OPACITY 2     /* opacities = sigmoid(raw) */
#define SR_ACT_NORMALIZE_ROTATIONS 4 /* rotations = raw / max(|raw|, 1e-12) */

typedef enum SrStatus {
    SR_OK = 0,
    SR_ERR_INVALID_ARGUMENT = -1, /* NULL where a pointer is required, bad sizes, both/neither of an exclusive pair */
    SR_ERR_HIP = -2,              /* a HIP runtime call or kernel launch failed */
    SR_ERR_BUFFER_TOO_SMALL = -3, /* a state buffer is smaller than sr_*_bytes() reports */
    SR_ERR_UNSUPPORTED = -4       /* e.g. sh_degree > 3 */
} SrStatus;

/* Per-call camera / raster settings == GaussianRasterizationSettings
 * (/root/reference/gaussian_renderer/__init__.py:39-52). */
typedef struct SrFrame {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;     /* active degree (0..3) */
    int32_t prefiltered;   /* always 0 at the reference's call sites; accepted, ignored */
    int32_t debug;         /* 1: synchronise + check after every kernel */
    const float* bg;          /* device [NC] (3, or 6 with SrGaussians.color_channels == 6) */
    const float* viewmatrix;  /* device [16] = world_view_transform (W2C^T), row-major */
    const float* projmatrix;  /* device [16] = full_proj_transform ((P*W2C)^T), row-major */
    const float* campos;      /* device [3] */
    int32_t tile_width;       /* 0 = 16 (the reference's BLOCK_X); BASELINE config 5 sweeps 8x8, 16x8, 16x16, 32x8, 32x16.  The 6- and
                               * 9-channel passes (SrGaussians.color_channels) and the per-class pass exist for every shape, the counter
                               * variant (blend_counters) for 16x16 only */
    int32_t tile_height;      /* 0 = 16 (BLOCK_Y).  Same shape in every call that shares the state buffers */
    uint32_t flags;           /* SR_FLAG_* bits; per call, nothing about a call is process-wide state */
    uint64_t* blend_counters; /* NULL, or device [16] u64 owned by the caller: selects the COUNTING variant of the forward blend (same
                               * results, slow), which adds to [0] list entries staged, [1] entries kept by the quadrant culling,
                               * [2] (entry, quadrant) tests run, [3] tests with >= 1 contributing pixel, [4] contributing (pixel, entry)
                               * pairs, [5] / [6] tests with a contributing pixel in rows 0-3 / rows 4-7 of the quadrant, [7] entries with a
                               * contributing pixel anywhere in the tile (= the gradient records the backward writes), [8] (entry, 4x4 cell) pairs
                               * with a contributing pixel, [9] the wave steps a 16-lane-row mapping would take (sum over rounds of 64
                               * entries and quadrants of the busiest cell's pair count), [10] / [11] the same two counting the pairs an octagon-vs-cell
                               * culling at staging would keep (hits and misses), [12] / [13] the same two with an octagon + oriented-box culling, [14] exact (entry, cell) hits that box would DROP (must be 0); [15] reserved.  16x16 tile with
                               * 3 or 6 colour channels; any other request returns SR_ERR_UNSUPPORTED (never silent zeros) */
} SrFrame;
#define SR_FLAG_NO_QUADRANT_CULL 1u  /* forward blend: run every list entry against every 8x8 quadrant instead of dropping entries that
                                      * provably cannot reach alpha >= 1/255 there.  Results are bit-identical either way (a test
                                      * requires it); the switch exists for that test and for A/B timing */
#define SR_FLAG_BALLOT_RANKING 2u    /* binning: rank the items of a wave with match-any ballots instead of LDS-atomic return values.  The
                                      * atomic path relies on gfx950 returning ds_add_rtn_u32 results in lane order (undocumented), so
                                      * the library checks that property on every device before its first sort (sr_rank_mode) and falls
                                      * back to the ballots by itself; this flag forces the fallback for one call (tests, A/B timing).
                                      * The lists are bit-identical either way */
#define SR_FLAG_ROW_MAPPED_FORWARD 4u /* forward blend, 16x16 tile with three colour channels: force the row-mapped kernel (the four 16-lane rows
                                      * of a wave are the four 4x4 cells of a quadrant, every row walks its own list of entries) ... */
#define SR_FLAG_QUADRANT_MAPPED_FORWARD 8u /* ... or the quadrant-mapped one (one entry on all 64 lanes).  Bit-identical images, state and hit
                                      * masks (a test requires it).  Without either flag the DEVICE picks per frame from the duplicates per
                                      * visible Gaussian the emission scan leaves in the geometry state: rows below 6.5 (small footprints:
                                      * -3 % at 1920x1080 / 3 M, -11 % at 1280x720), quadrants above (+10 % for the rows at 3840x2160):
                                      * DESIGN.md 4.  The flags exist for the A/B and the test.  SR_FLAG_ROW_MAPPED_FORWARD with any other
                                      * tile shape / channel count / blend_counters / SR_FLAG_NO_QUADRANT_CULL: SR_ERR_UNSUPPORTED */

#define SR_FLAG_FORWARD_ONLY 16u      /* no backward will follow this forward (the reference's inference callers run the operator under
                                      * torch.no_grad(): /root/reference/render.py:68, /root/reference/utils/mesh_utils.py:82-100): K1 does not
                                      * compute or write the 36-B/Gaussian SH direction Jacobian, K6 does not write the backward's state
                                      * (final_T / M1 / M2, last and median contributor: 20 B/pixel; the 2-B/duplicate hit masks).  Set in
                                      * BOTH sr_forward_plan and sr_forward_render; `image` may then be NULL / 0 bytes.  out_color, out_allmap
                                      * and radii are bit-identical to the training forward (a test requires it); calling a backward on the
                                      * state of such a forward is undefined */

#define SR_FLAG_NO_PRECOMP_COLOR_GRAD 32u /* backward (sr_backward / sr_backward_blend), 6 / 9 colour channels: the caller does not want dL/dcolors_precomp
                                      * (SrGradients.dL_dcolors NULL) -- the one-hot class channels of render_semantic are constants -- so K7 does not
                                      * form those sums (the channels still feed dL/dalpha); every other gradient is unchanged */

#define SR_FLAG_BINNING_CAPACITY 64u  /* the forward WITHOUT its host read-back (round 6).  By default sr_forward_plan waits once for D, the frame's duplicate
                                      * count, because the caller sizes the binning buffer from it -- as the reference does.  With this flag, set in EVERY call of
                                      * the frame (plan, render, backward):
                                      *   - sr_forward_plan queues K1, the scan and the depth sort and returns at once; *num_rendered_host = 0xFFFFFFFF;
                                      *   - `num_rendered` of sr_forward_render / sr_binning_bytes / sr_backward* / sr_backward_workspace_bytes is the CAPACITY (in
                                      *     duplicates) the caller chose for the binning buffer and the workspace -- e.g. 1.25 x the largest D it has seen;
                                      *   - a one-thread guard kernel compares the frame's D with the capacity on the device.  It fits: results bit-identical to
                                      *     the default mode.  It does not: nothing is binned, the images hold the background, every gradient is zero, no kernel reads or
                                      *     writes beyond the buffers, and SrGeomView.frame_counts[2] = 1 -- the caller looks at that word when it next synchronises
                                      *     anyway (frame_counts[0] = the exact D) and renders the frame again with a larger buffer.
                                      * No pinned memory, event or host wait is touched: the whole forward + backward can be captured into a HIP graph.
                                      * Not for the per-class passes (SR_ERR_UNSUPPORTED). */

#define SR_FLAG_ONE_SWEEP_SORT 128u   /* sr_forward_plan / sr_debug_radix_sort: the depth sort's four passes as ONE-SWEEP passes (one up-front digit count, then
                                      * one kernel per pass with decoupled look-back over relaxed-atomic status words: six launches instead of twelve).  Bit-identical
                                      * order; measured SLOWER on MI355X at the benchmark's size (0.190 vs 0.157 ms at 3 M keys: csrc/radix_sort.hip), hence a flag */

#define SR_FLAG_ONE_WAVE_BACKWARD 256u /* both blend kernels, 16x16 tile with three colour channels: force the full-size forms (two band waves per tile forward, one
                                      * wave per tile backward) ... */
#define SR_FLAG_COOP_BACKWARD 512u     /* ... or the cooperative ones (four waves per tile, one per 8x8 quadrant, ONE staging of every entry; backward: one record per
                                      * duplicate as before; forward: images and state bit-identical to the band kernel's -- measured no faster anywhere, so it
                                      * only runs with this flag).  Set in sr_forward_render and in the backward.  Without either
                                      * flag the library picks by the frame's tile count: cooperative below 2 600 tiles of 16x16 (the reference's `-r 4` frames:
                                      * 600 tiles cannot fill the GPU with one wave each), one wave per tile above.  Same tile lists either way; gradients
                                      * agree up to the order of a four-term sum.  The flags exist for the A/B and the tests */

#define SR_FLAG_ROW_BACKWARD 1024u     /* 16x16 tile, three colour channels, culling on: the ROW-MAPPED blend pair -- the forward (row-mapped kernel) writes its hit masks
                                      * per (entry, 4x4 cell) instead of per (entry, 8x8 quadrant), and the backward's four 16-lane rows each walk their own cell's
                                      * list (render_backward_rows_kernel).  Must be set in sr_forward_render AND in the backward of the same frame (the hit-mask
                                      * format differs).  Images bit-identical; gradients agree with the one-wave kernel's up to the order of additions.
                                      * MEASURED SLOWER (C3: K7 2.55 ms against 1.65: the per-entry sums are scattered into LDS with float atomics, ~100 LDS cycles
                                      * per wave instruction -- csrc/render_bwd.hip): built and kept as the measured answer, picked nowhere */

/* Per-Gaussian inputs == the keyword arguments of GaussianRasterizer.forward
 * (/root/reference/gaussian_renderer/__init__.py:129-138).  Exactly one of shs / colors_precomp
 * and exactly one of (scales, rotations) / transMat_precomp must be non-NULL. */
typedef struct SrGaussians {
    int32_t P;          /* number of Gaussians */
    int32_t sh_coeffs;  /* M = shs.size(1) (16 for max degree 3); 0 with colors_precomp */
    int32_t color_channels; /* NC: 0 or 3 = rgb; 6 = six precomputed channels (shs must be NULL); 9 = rgb from shs PLUS the six
                             * channels of colors_precomp[P,6] (both non-NULL): render + render_semantic as one pass */
    int32_t activations;    /* SR_ACT_* bits: the given arrays are the reference's RAW parameters and the activation is fused
                             * into K1 (and its adjoint into K8: raw gradients out); 0 = activated inputs as in the reference */
    const float* means3D;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* shs;
    const float* colors_precomp;
    const float* transMat_precomp;  /* the reference's `cov3D_precomp` slot carries a [P,9] transMat in 2DGS */
    const uint8_t* mask;            /* optional [P] (torch.bool storage): 0 = leave this Gaussian out, exactly as if the caller had
                                     * boolean-indexed every input (render_with_mask / semantic filters,
                                     * /root/reference/gaussian_renderer/__init__.py:89-105, 190-325) -- without the copies; radii
                                     * and gradients stay full-size, zero where masked out.  NULL = all */
} SrGaussians;

/* Gradient outputs of the backward == return tuple of _C.rasterize_gaussians_backward.
 * Every non-NULL pointer is fully written (zeros for invisible Gaussians); NULL = not wanted. */
typedef struct SrGradients {
    float* dL_dmeans2D;    /* [P,3] densification proxy (x, y, 0) -- what viewspace_points.grad receives */
    float* dL_dcolors;     /* [P,NC] w.r.t. colors_precomp ([P,6] for NC = 9); with shs as the colour source: [P,3] clamp-masked dL/drgb (the SH adjoint's input, see sr_sh_gradient_expand) */
    float* dL_dopacity;    /* [P,1] */
    float* dL_dmeans3D;    /* [P,3] */
    float* dL_dtransMat;   /* [P,9] */
    float* dL_dsh;         /* [P,M,3]; may be NULL with shs given (frame-parallel ranks ship dL_dcolors instead) */
    float* dL_dscales;     /* [P,2] */
    float* dL_drotations;  /* [P,4] */
} SrGradients;

/* Views into the caller-owned state buffers (for tests / debugging; all device pointers). */
typedef struct SrGeomView {
    const float* splats;           /* [P,20] packed record: Tu.xyz Tv.x | Tv.yz Tw.xy | Tw.z xy.x xy.y opacity | n.xyz r | g b view-depth radius.  Rows with radii == 0 are UNDEFINED: K1 does not write the
                                    * 128-B lines of the record array whose rows are all culled (round 5), so such rows hold whatever the caller's buffer held -- possibly
                                    * NaN; read a row only where radii > 0 (every kernel reaches the records through the tile lists, which hold visible Gaussians only) */
    const uint32_t* depth_keys;    /* [P] float bits of view-space depth; 0xFFFFFFFF when culled */
    const uint32_t* tiles_touched; /* [P] */
    const uint8_t* clamped;        /* [P] bit c set when SH colour channel c was clamped at 0 */
    const uint32_t* sorted_gid;    /* the frame_counts[1] visible Gaussians' ids in ascending (depth bits, id) order (room for P; culled ones are dropped) */
    const uint32_t* frame_counts;  /* [2] D (= *num_rendered_host of sr_forward_plan) and the number of Gaussians with at least one tile: what
                                    * the forward blend picks its mapping by (SR_FLAG_ROW_MAPPED_FORWARD).  With SR_FLAG_BINNING_CAPACITY: [3] -- after
                                    * sr_forward_render [2] = 1 if D exceeded the capacity (and [1] was zeroed: nothing was rendered), else 0 */
} SrGeomView;

typedef struct SrBinningView {
    const uint32_t* point_list;  /* [D] Gaussian id of every sorted duplicate (tile-major, then depth, then id); the tile of entry j is
                                  * the one whose range contains j (the tile ids themselves are not kept after the partition) */
    const uint32_t* ranges;      /* [tiles,2] (begin, end) into point_list; (0,0) for empty tiles */
    const uint32_t* tile_order;  /* [tiles] dispatch order of the blend waves: a permutation, longest list classes first */
} SrBinningView;

typedef struct SrImageView {
    const float* final_T;       /* [3,H,W]: T_final, M1, M2 */
    const uint32_t* n_contrib;  /* [2,H,W]: last_contributor, median_contributor */
} SrImageView;

int sr_abi_version(void);
/* Which of the named compile-time switches (include/surfel_switches.h: SURVEY.md Appendix A's (!) items) this library was built with
 * at a NON-default value: SR_SWITCH_BITS, 0 for the shipped configuration. */
uint32_t sr_build_switches(void);
/* The content digest of the sources this library was built from (csrc/*, include/*.h, the build script; 16 hex digits, "unknown" for a build
 * that did not pass one): the in-tree build rebuilds on a mismatch and the Python loader refuses a library that is not its tree's. */
const char* sr_source_digest(void);
const char* sr_last_error(void);

/* Sizes of the three state buffers (bytes). num_rendered = D from sr_forward_plan. */
size_t sr_geom_bytes(int32_t P);
size_t sr_binning_bytes(int32_t P, uint32_t num_rendered, int32_t image_width, int32_t image_height);
size_t sr_image_bytes(int32_t image_width, int32_t image_height);
size_t sr_backward_workspace_bytes(int32_t P, uint32_t num_rendered, int32_t color_channels);

int sr_geom_view(void* geom, size_t geom_bytes, int32_t P, SrGeomView* out);
int sr_binning_view(void* binning, size_t binning_bytes, int32_t P, uint32_t num_rendered, int32_t image_width,
                    int32_t image_height, SrBinningView* out);
int sr_image_view(void* image, size_t image_bytes, int32_t image_width, int32_t image_height, SrImageView* out);

/* Forward, phase 1 (K1 preprocess + tile-count scan + depth ordering).
 * Writes radii[P] (int32) and the geometry state; returns D = number of (tile, Gaussian) duplicates in
 * *num_rendered_host after ONE host wait (the same read-back the reference does between its scan and duplicateWithKeys;
 * the depth sort is queued behind the copy of D, so the GPU keeps working while the caller sizes the binning buffer). */
int sr_forward_plan(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, int32_t* radii,
                    uint32_t* num_rendered_host, void* stream);

/* Forward, phase 2 (K3 / K4 expanding tile partition: column pass over the Gaussians, row pass over the column items; K5 ranges + dispatch
 * order; K6 blend).
 * out_color [NC,H,W], out_allmap [7,H,W] (0 sum w*depth, 1 alpha, 2-4 sum w*normal (view space),
 * 5 median depth, 6 distortion). */
int sr_forward_render(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning,
                      size_t binning_bytes, void* image, size_t image_bytes, uint32_t num_rendered,
                      float* out_color, float* out_allmap, void* stream);

/* Backward (K7 blend backward + K8 preprocess backward).  dL_dcolor [NC,H,W], dL_dallmap [7,H,W].
 * workspace: sr_backward_workspace_bytes(P, num_rendered, NC) bytes (one 96-B -- 112-B for NC = 9 -- gradient record + one flag byte per
 * (tile, Gaussian) duplicate), contents undefined on entry. */
int sr_backward(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes,
                void* binning, size_t binning_bytes, void* image, size_t image_bytes, uint32_t num_rendered,
                const float* dL_dcolor, const float* dL_dallmap, void* workspace, size_t workspace_bytes,
                const SrGradients* grads, void* stream);

/* CONTRACT of every backward entry point (sr_backward, sr_backward_geometry, sr_class_backward): `frame` and the geometry inputs of `g`
 * (means3D, scales, rotations or transMat_precomp, activations, mask) must be BIT-IDENTICAL to those of the forward that filled the state
 * buffers.  K8 does not read Tu / Tv / Tw / the centre back from the 80-B record: it recomputes them from these inputs (the same device
 * functions as K1, hence the same bits) -- with other values the reference centre of K7's moment flush and K8's chain would silently
 * disagree.  The autograd shim saves and passes the forward's own tensors. */

/* The same backward in two halves, for callers that want to start a gradient exchange in between:
 *   sr_backward_blend     K7: per-(tile, Gaussian) gradient records into `workspace`;
 *   sr_backward_colors    optional, 3-channel pass: dL_dcolors[P,3] from those records alone (clamp-masked dL/drgb when shs is the
 *                         colour source = the input of sr_sh_gradient_expand; plain dL/dcolors_precomp otherwise) -- bit-identical
 *                         to what sr_backward_geometry returns in SrGradients.dL_dcolors;
 *   sr_backward_geometry  K8: everything else, from the same workspace.
 * sr_backward == sr_backward_blend + sr_backward_geometry. */
int sr_backward_blend(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                      void* image, size_t image_bytes, uint32_t num_rendered, const float* dL_dcolor, const float* dL_dallmap,
                      void* workspace, size_t workspace_bytes, void* stream);
int sr_backward_colors(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes,
                       uint32_t num_rendered, void* workspace, size_t workspace_bytes, float* dL_dcolors, void* stream);
int sr_backward_geometry(const SrFrame* frame, const SrGaussians* g, const int32_t* radii, void* geom, size_t geom_bytes,
                         void* binning, size_t binning_bytes, void* image, size_t image_bytes, uint32_t num_rendered, void* workspace,
                         size_t workspace_bytes, const SrGradients* grads, void* stream);

/* Per-class distortion pass.  The reference's training iteration renders the same view once per semantic class
 * with the other classes boolean-indexed away and keeps only `rend_dist` of each (/root/reference/train.py:94-103 ->
 * gaussian_renderer/__init__.py:89-105, 165): five full rasterizations for five distortion maps.  This pass runs K1..K5 once and
 * returns all maps: out_dist[k] == allmap[6] of the operator called on the class-k subset (same list order, same thresholds, same
 * early termination per class), and the backward returns the gradients of all of them together.
 *   - class ids: colors_precomp[P,3], column 0 holds the class of a Gaussian as a float (0 .. n_classes-1; negative or >= n_classes =
 *     in no class, contributes nowhere); shs must be NULL.  sr_forward_plan is called first, exactly as for a render.
 *   - class_image: sr_class_image_bytes(W, H, n_classes) bytes of caller-owned state between forward and backward.
 *   - out_dist / dL_ddist: [n_classes, H, W].  grads: as sr_backward (dL_dcolors / dL_dsh are not produced: pass NULL).
 *   - every tile shape of the sweep; 1 <= n_classes <= 6. */
size_t sr_class_image_bytes(int32_t image_width, int32_t image_height, int32_t n_classes);
int sr_class_forward_render(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, void* geom, size_t geom_bytes, void* binning,
                            size_t binning_bytes, void* class_image, size_t class_image_bytes, uint32_t num_rendered, float* out_dist,
                            void* stream);
int sr_class_backward(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, const int32_t* radii, void* geom, size_t geom_bytes,
                      void* binning, size_t binning_bytes, void* class_image, size_t class_image_bytes, uint32_t num_rendered,
                      const float* dL_ddist, void* workspace, size_t workspace_bytes, const SrGradients* grads, void* stream);

/* The same per-class pass on the plan AND the binning of a colour pass of the same frame (SURVEY.md 8f N1 in full: K1..K5 once for the
 * colour render, the semantic channels and the per-class distortion maps; one K8 for all of their gradients):
 *   forward:  sr_forward_plan -> sr_forward_render (any colour configuration) -> sr_class_forward_shared
 *   backward: sr_backward_blend -> sr_class_backward_shared -> sr_backward_geometry
 * `classes` [P] int32 (negative or >= n_classes: in no class).  class_state: sr_class_shared_bytes(P, W, H, n_classes, num_rendered) bytes of
 * caller-owned state between forward and backward (per-class image state, class bytes, this pass's own hit masks; the class-ordered list
 * goes into a region of `binning` the colour pass no longer needs).  sr_class_backward_shared ADDS its gradient sums to the records
 * sr_backward_blend left in `workspace` (and starts records for duplicates only the class chains reached), so the one
 * sr_backward_geometry that follows returns the gradients of colour, allmap AND distortion maps together.  `g` / `frame` as given to
 * the colour pass (scales + rotations, no precomputed transMat). */
size_t sr_class_shared_bytes(int32_t P, int32_t image_width, int32_t image_height, int32_t n_classes, uint32_t num_rendered);
int sr_class_forward_shared(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, const int32_t* classes, void* geom, size_t geom_bytes,
                            void* binning, size_t binning_bytes, void* class_state, size_t class_state_bytes, uint32_t num_rendered,
                            float* out_dist, void* stream);
int sr_class_backward_shared(const SrFrame* frame, const SrGaussians* g, int32_t n_classes, void* geom, size_t geom_bytes, void* binning,
                             size_t binning_bytes, void* class_state, size_t class_state_bytes, uint32_t num_rendered, const float* dL_ddist,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Frame-parallel SH gradient (no reference counterpart -- the reference is single-GPU).  The SH adjoint is
 * linear in the clamp-masked colour gradient and its only other per-view input is the camera position, so ranks that
 * rendered n_views frames of the SAME Gaussians all-gather dL_dcolors (12 B/Gaussian) instead of all-reducing dL_dsh
 * (192 B/Gaussian) and each expands the sum:  dL_dsh[i][k][c] = sum_v basis_k(dir(means3D[i], campos[v])) * dL_dcolors[v][i][c].
 * campos [n_views,3], dL_dcolors [n_views,P,3], dL_dsh [P,M,3] (fully written), all device pointers.  n_views = 1
 * reproduces sr_backward's dL_dsh bit for bit. */
int sr_sh_gradient_expand(int32_t P, int32_t sh_coeffs, int32_t sh_degree, int32_t n_views, const float* means3D,
                          const float* campos, const float* dL_dcolors, float* dL_dsh, void* stream);

/* K-nearest-neighbour mean squared distance == simple_knn._C.dist3knn / dist10knn (K = 3 / 10; scale
 * initialisation, /root/reference/scene/gaussian_model.py:16,151) and meanDistFromReferencePcd (distance of every point of
 * one cloud to another cloud, /root/reference/inpainting_pipeline/2_condition_preparation/2_generate_inpainted_mask.py:27,71-73).
 * out[i] = (sum of the K smallest squared distances from query i to the reference points) / K; with query == NULL every
 * reference point is searched against the OTHER reference points (n_query ignored, out [n_reference]).  Exact search.
 * points are [n,3] float32 device arrays; workspace: sr_knn_workspace_bytes(n_query or 0, n_reference) bytes. */
size_t sr_knn_workspace_bytes(int32_t n_query, int32_t n_reference);
int sr_knn_mean_dist2(int32_t n_query, const float* query, int32_t n_reference, const float* reference, int32_t K,
                      int32_t take_sqrt, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* K9: present[i] = (view-space z of means3D[i] > 0.2).  present is uint8 (torch.bool storage). */
int sr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, void* stream);

/* Fused post-processing of the operator's allmap == gaussian_renderer/__init__.py:152-177 +
 * utils/point_utils.py:9-37 of the reference: allmap[7,H,W] -> rend_normal[3,H,W] (world space), surf_depth[1,H,W],
 * surf_normal[3,H,W] (finite-difference pseudo-normals times alpha, zero border), surf_point[3,H,W].
 * fovx/fovy in radians, viewmatrix = world_view_transform (device [16]).  Backward: gradient w.r.t. allmap from the
 * gradients of the four outputs (NULL = zero); scratch6 is [6,H,W] floats of caller-owned scratch; channel 6 of
 * g_allmap is written as 0 and alpha is treated as detached inside surf_normal, as in the reference. */
int sr_postprocess_forward(int32_t image_width, int32_t image_height, float fovx, float fovy, float depth_ratio,
                           const float* viewmatrix, const float* allmap, float* rend_normal, float* surf_depth,
                           float* surf_normal, float* surf_point, void* stream);
int sr_postprocess_backward(int32_t image_width, int32_t image_height, float fovx, float fovy, float depth_ratio,
                            const float* viewmatrix, const float* allmap, const float* g_rend_normal, const float* g_surf_depth,
                            const float* g_surf_normal, const float* g_surf_point, float* scratch6, float* g_allmap, void* stream);

/* Test hook of the parity bars: the hard decisions the blend kernels take, dumped per (list entry, pixel) pair.  For list position
 * j (index into SrBinningView.point_list) and 8x8 quadrant q of its tile (q = (y / 8) * (tile_width / 8) + x / 8, bit = (y % 8) * 8 + x % 8
 * in tile-local pixel coordinates): valid_bits[j * nq + q] = pixels where the entry passes the chain of skips of the forward blend
 * (p.z != 0, depth >= near, power <= 0, alpha >= 1/255 -- NOT the pixel's saturation state), use3d_bits = pixels where it takes the
 * ray-splat path (rho3d <= rho2d).  Computed by the same device functions on the same staged values as K6 / K7, i.e. these are the
 * bits they act on; tests/ hands them to the CPU oracle so that both sides blend the same contributor sets.
 * Both arrays: device [num_rendered * nq] u64, nq = (tile_width / 8) * (tile_height / 8). */
int sr_debug_pair_decisions(const SrFrame* frame, const SrGaussians* g, void* geom, size_t geom_bytes, void* binning,
                            size_t binning_bytes, uint32_t num_rendered, uint64_t* valid_bits, uint64_t* use3d_bits, void* stream);

/* Test hook for the library's stable LSD radix sort (binning K2/K4): sorts n (key, value) u32 pairs by key bits
 * [0, total_bits); vals_in == NULL means value = index.  temp: sr_debug_radix_sort_temp_bytes(n) bytes. */
size_t sr_debug_radix_sort_temp_bytes(uint32_t n);
int sr_debug_radix_sort(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t n,
                        int total_bits, void* temp, size_t temp_bytes, uint32_t flags /* SR_FLAG_BALLOT_RANKING or 0 */, void* stream);

/* How the sort / partition kernels of the current device rank items (the run-time guard of the LDS-atomic lane-order assumption):
 * runs the self-check on first use (one tiny kernel + one stream wait per device and process), then returns the cached answer:
 * 1 = LDS-atomic return values (the property holds), 2 = match-any ballots (it does not: the fallback is in use), or
 * SR_ERR_UNSUPPORTED if neither ranking reproduces the reference ranks on this device. */
int sr_rank_mode(void* stream);

/* Test hook for the hardware property the sort kernels rank by: within one wave instruction, ds_add_rtn_u32 returns the old values
 * to the lanes that hit the same LDS address in ascending lane order.  ranks[i] = what lane i % 64 of its wave got back when every
 * lane added 1 to counter digits[i] % bins of its wave (n a multiple of 256, processed 256 per block step, 8 steps per block, the
 * counters running on across the steps; bins <= 1024).  All device pointers. */
int sr_debug_lds_atomic_ranks(const uint32_t* digits, uint32_t* ranks, uint32_t n, int bins, void* stream);

/* Profiling aid -- the one piece of process-wide state in this library (mutex-protected: autograd runs the backward on its own
 * thread).  sr_set_stage_timing(1) makes every later call bracket each stage with a pair of HIP events recorded
 * on the caller's stream (no host sync while recording; up to 512 launches per stage); sr_set_stage_timing(2 * mask),
 * mask = OR of (1 << SrStage), brackets only the stages in the mask (every event pair costs a few microseconds of
 * stream time: the full set adds ~1.4 % to a 4.5 ms step).  sr_stage_stats() waits for the recorded events and returns the
 * summed duration (ms) and the number of launches of one stage since timing was (re-)enabled. */
typedef enum SrStage {
    SR_STAGE_PREPROCESS = 0, SR_STAGE_DEPTH_SORT = 1, SR_STAGE_SCAN = 2, SR_STAGE_EXPAND_X = 3, SR_STAGE_EXPAND_Y = 4,
    SR_STAGE_RANGES = 5, SR_STAGE_BLEND_FWD = 6, SR_STAGE_BLEND_BWD = 7, SR_STAGE_PREPROCESS_BWD = 8,
    /* the per-class distortion pass's own kernels (its K1..K5 and K8 are the stages above) */
    SR_STAGE_CLASS_PARTITION = 9, SR_STAGE_CLASS_FWD = 10, SR_STAGE_CLASS_BWD = 11, SR_STAGE_COUNT = 12
} SrStage;
void sr_set_stage_timing(int enable);
int sr_stage_stats(int stage, float* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* SURFEL_RASTER_H */
