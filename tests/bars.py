"""THE table of parity bars (round 6: frozen).

Every tolerance the parity tests apply lives here, once, with the commit that last changed it.  tests/gpu_util.py and the full-size
tests read their constants from `BARS`; nothing else in tests/ may define a tolerance of its own for the operator's outputs.
`tests/golden/bars_frozen.json` is the snapshot of the values at the freeze; `tests/test_bars_frozen.py` fails if

  * a bar here is LOOSER (larger) than its frozen value, or a frozen bar disappeared, unless DESIGN.md section 3 carries a line
    `BARS_CHANGELOG: <name> <old> -> <new> ...` naming it (with the reference-side argument for it);
  * a `# bar:` literal creeps back into gpu_util.py / test_gpu_fullsize.py outside this table.

Tightening a bar needs no entry.  The five widenings of round 5 (the review listed them) are marked `widened r5`.

Reading the bars: north_star's "within 1e-4" holds for ROBUST pixels only -- pixels none of whose decisions (contribute, path, stop,
median) sits within float32 noise of its threshold under the free-running float64 checker.  Say "robust pixels", never "within 1e-4".
"""

# name: (value, last changed in, what it bounds)
BARS = {
    # ---- images --------------------------------------------------------------------------------------------------------------
    "robust_pixel":            (1e-4,  "3c07aaf", "|kernels - float64| / (1 + |v|), every colour / aux-map element with identical decisions, and every robust pixel free-running"),
    "nonrobust_pixel_cap":     (2e-2,  "6a7bf5e", "the same metric at a non-robust pixel, small scenes (a flipped contributor moves a pixel by about alpha)"),
    "nonrobust_pixel_cap_full": (6e-3, "3785f10", "... at the full-size configurations (measured 2.1e-3 at C2 / C3 / clustered: three times that)"),
    "elementwise_over_1e4_frac": (1e-6, "25bc64f", "fraction of image elements that may exceed 1e-4 where the float32 oracle is worse at that very element (full-size general pose: 1 of 20.7 M)"),
    "oracle32_image_atol":     (1e-4,  "initial", "float32-oracle comparison: atol and rtol of the fraction bar"),
    "oracle32_image_bad_frac": (2e-4,  "initial", "... fraction of elements that may miss it, full size"),
    "oracle32_image_bad_frac_small": (5e-4, "initial", "... on the small scenes (check_allmap default)"),
    "oracle32_image_hard":     (2e-2,  "initial", "... no element further than this times the tensor scale"),
    # ---- gradient rows ---------------------------------------------------------------------------------------------------------
    "row_p999":                (2e-4,  "3c07aaf", "99.9 % of the visible rows of every gradient tensor, relative to the row (scales / rotations / means3D: to the terms K8 sums)"),
    "row_p999_means2D":        (6e-4,  "3c07aaf", "... of the densification proxy dL_dmeans2D (one cancelling float32 sum, not a row maximum)"),
    "row_max":                 (1e-2,  "3c07aaf", "every row"),
    "row_p999_fraction":       (1e-3,  "3c07aaf", "the fraction of rows that may exceed row_p999 (what makes it a 99.9th percentile; two rows in any case)"),
    "row_floor":               (1e-3,  "3c07aaf", "a row's error is relative to max(|row|) + this x max(|tensor|) (rows that are zero in the reference)"),
    "row_outlier_factor":      (5.0,   "805cb52", "widened r5: with outlier_frac > 0 (C4 cameras only) that fraction of rows may reach this x row_max (1e-2 -> 5e-2 for one row in a million)"),
    "row_outlier_frac_c4":     (1e-6,  "805cb52", "widened r5: the outlier fraction the C4 camera tests pass"),
    "row_plain_p999":          (2e-3,  "3785f10", "dL_dscales / dL_drotations rows under the plain row metric (no scene: precomputed transMat)"),
    "row_plain_max":           (6e-2,  "3785f10", "... every row"),
    "row_oracle32_excuse":     (2.0,   "a685fe6", "widened r5: a row above row_max is accepted where the float32 oracle's own error on that row is at least 1 / this of it"),
    "nonrobust_row_cap":       (5e-2,  "6a7bf5e", "a non-robust row, of the tensor scale, small scenes"),
    "nonrobust_row_cap_full":  (5e-3,  "3785f10", "... at the full-size configurations (measured 1.5e-3)"),
    "oracle32_grad_rel":       (2e-3,  "initial", "float32-oracle comparison of the gradients: all but oracle32_grad_bad_frac of the elements within this of the tensor scale"),
    "oracle32_grad_bad_frac":  (1e-3,  "initial", "... that fraction"),
    "oracle32_grad_hard":      (5e-2,  "initial", "... every element, small scenes"),
    "oracle32_grad_hard_full": (0.25,  "25bc64f", "... at full size (the float32 ORACLE's worst rows are 0.3 of a row off float64 there)"),
    "oracle32_grad_hard_posed": (0.5,  "25bc64f", "widened r5: ... for the full-size general-pose scene (the float32 oracle's worst row there: 0.35 of the tensor scale off float64, the kernels' 2.6e-3)"),
    # ---- counted budgets ---------------------------------------------------------------------------------------------------------
    "nonrobust_pixel_budget":  (1.5e-2, "6cfb9c9", "fraction of pixels that may be non-robust, small scenes"),
    "nonrobust_gaussian_budget": (0.40, "6cfb9c9", "fraction of visible Gaussians with a near-threshold decision somewhere in their footprint, small scenes"),
    "nonrobust_pixel_budget_c2": (2.5e-3, "6cfb9c9", "... C2 (measured 0.17 %)"),
    "nonrobust_pixel_budget_c3": (8e-3, "6cfb9c9", "... C3 / C4 (measured 0.57 %)"),
    "nonrobust_gaussian_budget_full": (0.25, "6cfb9c9", "... visible Gaussians at full size (measured 19-21 %)"),
    "nonrobust_pixel_budget_c5": (6e-3, "initial", "... the C5 scene (6 M at 3840x2160)"),
    "nonrobust_gaussian_budget_c5": (0.20, "initial", "... its visible Gaussians"),
    "nonrobust_pixel_budget_clustered": (1e-2, "initial", "... the clustered (street-like, heavy-tailed) scene"),
    "nonrobust_gaussian_budget_clustered": (0.26, "initial", "... its visible Gaussians"),
    "differing_pixel_frac":    (1e-4,  "f764ce6", "fraction of the frame that may hold a decision differing from the float64 checker's (all non-robust; C3: 1.5e-5)"),
    "differing_pixel_frac_fuzz_big": (5e-4, "f764ce6", "widened r5: ... in the FUZZ_BIG sweep of tools/fuzz_parity.py only (translucent regime, pixels thousands deep: measured 1.3e-4 .. 2.3e-4)"),
    "differing_of_nonrobust":  (5e-3,  "a685fe6", "widened r5: ... or, on a small frame, this fraction of the NON-ROBUST pixels (min 3 pixels)"),
    "n_contrib_mismatch_frac": (1e-3,  "initial", "fraction of pixels whose last / median contributor differs from the float32 oracle's, full size"),
    "value_range_slack":       (1e-6,  "initial", "range properties at full size: alpha <= 1 - T_stop + this, clamped colours >= -this"),
    # ---- extensions (class pass, one-plan pass) ------------------------------------------------------------------------------------
    "class_maps_vs_operator":  (1e-6,  "initial", "HIP vs HIP: distortion maps of the class pass against the operator on the class subsets, of max(1, |map|) (bit-identical in practice)"),
    "class_grads_vs_operator": (2e-5,  "initial", "HIP vs HIP: summed gradients of the class pass against the subset renders, of the tensor scale (summation order only)"),
    "class_pass_oracle32_factor": (1.25, "r6", "new r6: class-pass gradient rows vs the float64 oracle (tests/test_gpu_class_pass_oracle.py): p99.9 within row_p999, or within this x the float32 oracle's p99.9 on the same rows.  A distortion-only loss is a variance along the ray: upstream's float32 formulation itself sits at 1e-3 there, and the kernels are statistically indistinguishable from it (measured kernels / oracle p99.9 over 5 tensors x 3 tile shapes: 0.74 .. 1.09; the p99.9 of 14 k rows is its 14th largest)"),
    "extension_f64_fallback":  (3.0,   "9634a62", "widened r5: class-pass gradient vs the float64 backward may be this x the subset renders' own distance (2x -> 3x, seed 30703: 2.09x on rounding noise)"),
}


def bar(name):
    return BARS[name][0]
