"""GPU (-m gpu): the strict parity bar.  Both sides take the SAME hard decisions -- the HIP kernels' own (which pairs pass
alpha >= 1/255 / depth / power, ray-splat vs screen-space path, where each pixel stops at T < 1e-4, which contributor is the median:
sr_debug_pair_decisions + n_contrib) -- and the checker evaluates the blend in double precision on the same float32 per-Gaussian
state (oracle/surfel_blend.inc).  Then images must agree to north_star's 1e-4 at EVERY pixel and the gradients per Gaussian row,
relative to the row's own magnitude.  The free-running float32 oracle comparisons (tests/test_gpu_parity.py) keep their wider
bars: their excess is (a) decisions taken within rounding noise of a threshold and (b) the float32 oracle's own conditioning
(global pixel coordinates), both quantified in profiles/r02_parity.json."""
import json
import os

import numpy as np
import pytest
import torch

from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads

pytestmark = pytest.mark.gpu

SCENES = [  # P, W, H, seed, scale_lo, scale_hi, camera index, SH degree, tile
    (60000, 480, 270, 0, 5e-4, 5e-3, None, 3, None),       # BASELINE config 2's density and splat sizes, cropped
    (8000, 200, 150, 7, 5e-3, 8e-2, 6, 3, None),           # large splats, oblique camera
    (3000, 160, 96, 3, 2e-2, 3e-1, 1, 1, None),            # huge splats, long lists, SH degree 1
    (20000, 203, 117, 11, 2e-3, 3e-2, 2, 2, (32, 16)),     # ragged image, 32x16 tile
    (20000, 203, 117, 11, 2e-3, 3e-2, 2, 0, (8, 8)),       # 8x8 tile, SH degree 0
]


@pytest.mark.parametrize("scene", SCENES)
def test_strict_parity_with_identical_decisions(scene):
    from tests.gpu_util import assert_strict_parity, forced_f64_reference, run_hip
    P, W, H, seed, lo, hi, idx, deg, tile = scene
    cam = synthetic_camera(W, H) if idx is None else synthetic_camera(W, H, index=idx)
    g = synthetic_gaussians(P, W, H, seed=seed, scale_lo=lo, scale_hi=hi)
    g["opacities"][::9] = 1.0
    dc, da = synthetic_upstream_grads(W, H, seed=seed + 1)
    bg = [0.3, 0.1, 0.6]
    hip = run_hip(g, cam, bg, deg, dc, da, tile=tile)
    raw, fwd64, bwd64 = forced_f64_reference(g, cam, bg, deg, dc, da, tile=tile)
    np.testing.assert_array_equal(hip["radii"], fwd64["radii"])
    np.testing.assert_array_equal(raw["color"], hip["color"])          # the decision dump describes this very forward
    report = {}
    try:
        assert_strict_parity(hip, fwd64, bwd64, report=report, scene=(g, cam))
    finally:
        out = os.environ.get("SR_PARITY_REPORT")
        if out:
            with open(out, "a") as f:
                f.write(json.dumps({"scene": list(map(str, scene)), **report}) + "\n")


def test_strict_parity_precomputed_colours_and_empty_tiles():
    """colors_precomp as the colour source; a scene that leaves most tiles empty and one that ends lists early (opaque wall)."""
    from tests.gpu_util import assert_strict_parity, forced_f64_reference, run_hip
    W, H, P = 176, 112, 2500
    cam = synthetic_camera(W, H, index=4)
    g = synthetic_gaussians(P, W, H, seed=21, scale_lo=4e-3, scale_hi=4e-2)
    g["means3D"][:, 0] = g["means3D"][:, 0].abs() * 0.3           # everything in the right third of the frustum
    g["opacities"][:400] = 0.995                                   # saturating front layer
    colors = torch.rand(P, 3, generator=torch.Generator().manual_seed(2)).numpy()
    dc, da = synthetic_upstream_grads(W, H, seed=5)
    hip = run_hip(g, cam, [0, 0, 0], 0, dc, da, colors=colors)
    raw, fwd64, bwd64 = forced_f64_reference(g, cam, [0, 0, 0], 0, dc, da, colors=colors)
    assert_strict_parity(hip, fwd64, bwd64, scene=(g, cam))
