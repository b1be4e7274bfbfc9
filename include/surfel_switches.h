/*
 * surfel_switches.h -- the named compile-time switches of the surfel rasterizer: ONE list, included by the HIP kernels
 * (streetunveiler_amd/csrc/common.h) AND by the CPU oracle (oracle/surfel_oracle.c), so that the two are always configured alike.
 *
 * Why they exist.  The reference's native rasterizer is an un-vendored submodule (/root/reference/.gitmodules:9-12: a fork of
 * hbb1/diff-surfel-rasterization at tags/StreetUnveiler, no SHA), so the behavioural contract is SURVEY.md Appendix A, and the
 * items that appendix marks (!) are the ones upstream revisions / the fork may disagree on.  Each of them is a macro here; a
 * maintainer who holds the real CUDA source flips the one that differs --
 *     SR_EXTRA_HIPCC_FLAGS="-DSR_DETACH_WEIGHT=1" python -m streetunveiler_amd.build --force      (kernels)
 *     make -C oracle -B CFLAGS_EXTRA="-DSR_DETACH_WEIGHT=1"                                          (oracle)
 * or `python -m streetunveiler_amd.build --variant detach_weight` (both at once, into lib/variants/) -- without editing a kernel.
 * tests/test_gpu_switches.py builds every non-default value and checks kernels against oracle under it.
 * sr_build_switches() / so_build_switches() report what a loaded library was compiled with.
 */
#ifndef SURFEL_SWITCHES_H
#define SURFEL_SWITCHES_H

/* Upstream config.h `TIGHTBBOX` (0 in the paper's build): 1 = the bounding-box cutoff follows the opacity,
 * cutoff = sqrt(max(9 + 2 ln(opacity), 1e-6)) instead of 3 -- in the AABB of K1 and in the radius floor below; the AABB-centre
 * term of K8 keeps upstream's hard-coded t = (9, 9, -1).  Appendix A.0 / A.2 step 5. */
#ifndef SR_TIGHTBBOX
#define SR_TIGHTBBOX 0
#endif

/* Upstream config.h `DETACH_WEIGHT` (0): 1 = the distortion loss does not differentiate through the blend weights
 * (dL_dweight += 0 instead of (M2_final + m^2 A_final - 2 m M1_final) g_dist); the gradient through the depth metric m stays.
 * Appendix A.0 / A.5. */
#ifndef SR_DETACH_WEIGHT
#define SR_DETACH_WEIGHT 0
#endif

/* radius = ceil(max(extent.x, extent.y, cutoff * FilterSize)) (1, current upstream) or ceil(max(extent.x, extent.y)) (0, older
 * revisions).  Appendix A.2 step 6. */
#ifndef SR_RADIUS_FILTER_FLOOR
#define SR_RADIUS_FILTER_FLOOR 1
#endif

/* The median-depth gradient goes to the entry whose 0-based list index equals median_contributor - 1 (1: the forward stores the
 * 1-based position of the median entry, so this is that entry) or median_contributor (0: the indexing some revisions use -- the
 * entry BEHIND the median one).  Appendix A.5. */
#ifndef SR_MEDIAN_CONTRIBUTOR_MINUS_ONE
#define SR_MEDIAN_CONTRIBUTOR_MINUS_ONE 1
#endif

/* Densification proxy dL_dmean2D = (dL/dTu.z, dL/dTv.z) * depth_c * (W, H) / 2: depth_c = Tw.z = transMat[8] (0, upstream) or the
 * view-space depth of the centre (1).  Equal up to rounding under the reference's projection (clip w = view z); different with a
 * precomputed transMat.  Appendix A.6. */
#ifndef SR_PROXY_DEPTH_VIEW_Z
#define SR_PROXY_DEPTH_VIEW_Z 0
#endif

/* Image size inside the per-Gaussian backward (viewport of the dL/dT chain, and the proxy above): upstream's
 * int(focal * tanfov * 2) with focal = size / (2 tanfov), all in float32 (1, upstream: for some (size, fov) pairs the product lands
 * just below the integer and truncates to size - 1 -- the proxy gradient that drives densification is then scaled by (W - 1) / W),
 * or the operator's image_width / image_height (0: what exact arithmetic gives).  Appendix A.6 / DESIGN.md 3. */
#ifndef SR_BACKWARD_WH_FROM_FOCAL
#define SR_BACKWARD_WH_FROM_FOCAL 1
#endif

/* Upstream's per-pair `if (p.z == 0) continue` in the forward and backward blend (1, upstream: Appendix A.4), or (0) the
 * exact-arithmetic rule: a pair whose p.z vanishes is blended through its 2-D filter footprint (rho3d = inf), and a splat whose p.z
 * vanishes at EVERY pixel (a zero scale) contributes nowhere.  For a healthy splat p.z == 0 is rounding noise of the cross product
 * -- which pairs land on exactly 0 in float32 depends on the operation order, so with 1 the kernels evaluate the test on their own
 * staged cross product and the oracle on k x l (the margin walk of the oracle flags every pair within that noise as non-robust;
 * profiles/r05_parity_c3.json counts them).  Appendix A.4 / INTEGRATION.md "Differences". */
#ifndef SR_REFERENCE_PZ_SKIP
#define SR_REFERENCE_PZ_SKIP 1
#endif

/* one bit per switch that is NOT at its default; every default is upstream's behaviour as SURVEY.md Appendix A states it, so
 * sr_build_switches() == 0 / so_build_switches() == 0 reads "upstream semantics" */
#define SR_SWITCH_BITS ((SR_TIGHTBBOX ? 1u : 0u) | (SR_DETACH_WEIGHT ? 2u : 0u) | (SR_RADIUS_FILTER_FLOOR ? 0u : 4u) | \
                        (SR_MEDIAN_CONTRIBUTOR_MINUS_ONE ? 0u : 8u) | (SR_PROXY_DEPTH_VIEW_Z ? 16u : 0u) | \
                        (SR_BACKWARD_WH_FROM_FOCAL ? 0u : 32u) | (SR_REFERENCE_PZ_SKIP ? 0u : 64u))

#endif /* SURFEL_SWITCHES_H */
