"""GPU (-m gpu): BASELINE.json's full-size configurations.

C2 (500k Gaussians, 1920x1080, SH 3, fwd+bwd) is still small enough for the CPU oracle on the GPU box's host
(seconds with OpenMP), so it gets the full oracle comparison.  C3 (3 M Gaussians, all aux gradients) and a
3840x2160 frame are checked through size-independent properties: sortedness / consistency of the tile lists,
conservation of the duplicate count, determinism (bit-identical reruns -- there are no atomics), linearity of the
backward in the upstream gradients, and value ranges."""
import numpy as np
import pytest
import torch

from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


def test_c2_500k_against_oracle():
    from tests.gpu_util import assert_close_frac, assert_grads_close, run_hip, run_hip_raw, run_oracle
    P = 500_000
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=0)
    bg = np.zeros(3, np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=1, aux=False)      # C2: colour + alpha gradients only
    fwd, bwd = run_oracle(g, cam, bg, 3, dc, da)
    raw = run_hip_raw(g, cam, bg, 3)
    assert raw["D"] == fwd["num_rendered"]
    np.testing.assert_array_equal(raw["radii"], fwd["radii"])
    np.testing.assert_array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    assert (raw["img"]["n_contrib"].view(np.uint32) != fwd["n_contrib"]).mean() < 1e-3
    out = run_hip(g, cam, bg, 3, dc, da)
    assert_close_frac(out["color"], fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, "C2 color")
    from tests.gpu_util import check_allmap
    check_allmap(out["allmap"], fwd["allmap"], "C2")
    for k in ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
        assert_grads_close(out[k], bwd[k], 2e-3, "C2 " + k)
    # ... and the strict bar at the full C2 size: identical decisions, float64 arbiter (tests/test_gpu_strict_parity.py)
    from tests.gpu_util import assert_strict_parity, forced_f64_reference
    _, fwd64, bwd64 = forced_f64_reference(g, cam, bg, 3, dc, da)
    assert_strict_parity(out, fwd64, bwd64, tag="C2 ")


def _properties(P, W, H, check_linearity=True):
    from diff_surfel_rasterization import _C
    from tests.gpu_util import DEV, run_hip, settings_for
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=0)
    s = settings_for(cam, [0, 0, 0], 3)
    e = torch.empty(0, device=DEV)
    d = {k: v.to(DEV) for k, v in g.items()}
    D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
        s.bg, d["means3D"], e, d["opacities"], d["scales"], d["rotations"], 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
        s.tanfovy, H, W, d["shs"], 3, s.campos, False, False)
    gv, bv = _C.geom_view(geom, P), _C.binning_view(binning, P, D, W, H)
    tiles_touched = gv["tiles_touched"].long()
    assert int(tiles_touched.sum()) == D                                   # every duplicate emitted exactly once
    assert ((radii > 0) == (tiles_touched > 0)).all()
    tile_keys, pl, ranges = bv["tile_keys"].long(), bv["point_list"].long(), bv["ranges"].long()
    assert (tile_keys[1:] >= tile_keys[:-1]).all()                         # tile-major
    depth = gv["depth_keys"].long() & 0xFFFFFFFF
    key64 = tile_keys * (1 << 32) + depth[pl]
    assert (key64[1:] >= key64[:-1]).all()                                 # depth-sorted inside each tile
    ties = key64[1:] == key64[:-1]
    assert (pl[1:][ties] > pl[:-1][ties]).all()                            # stable: ties keep ascending Gaussian id
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.sum()) == D and (lens >= 0).all()
    nz = lens > 0
    # (the tile ids are rebuilt from the ranges: the partition keeps only the permutation)  every Gaussian appears exactly once in
    # every tile of its rectangle: as often as K1 counted, and only in tiles its centre +- radius box reaches
    assert torch.equal(torch.bincount(pl, minlength=P), tiles_touched)
    rec = gv["splats"]
    cx, cy, rad = rec[pl, 9], rec[pl, 10], rec[pl, 19]
    gx = (W + 15) // 16
    tx, ty = (tile_keys % gx).float(), (tile_keys // gx).float()
    assert ((cx + rad + 15 >= tx * 16) & (cx - rad < (tx + 1) * 16) & (cy + rad + 15 >= ty * 16) & (cy - rad < (ty + 1) * 16)).all()
    assert torch.isfinite(color).all() and torch.isfinite(allmap).all()
    alpha = allmap[1]
    assert (alpha >= 0).all() and (alpha <= 1 - 1e-4 + 1e-6).all()        # 1 - T with T never below the 1e-4 stop
    assert (color >= -1e-6).all()                                          # clamped colours, bg 0
    nc = _C.image_view(img, W, H)["n_contrib"].long()
    assert (nc[0] <= lens.view((H + 15) // 16, (W + 15) // 16).repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]).all()
    del geom, binning, img
    # determinism and linearity of the backward
    dc, da = synthetic_upstream_grads(W, H, seed=1)
    a = run_hip(g, cam, [0, 0, 0], 3, dc, da)
    b = run_hip(g, cam, [0, 0, 0], 3, dc, da)
    for k in ["color", "allmap", "dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} not deterministic")
    if check_linearity:
        c = run_hip(g, cam, [0, 0, 0], 3, dc * 2, da * 2)   # power-of-two scaling is exact in float
        for k in ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
            np.testing.assert_array_equal(c[k], 2 * a[k], err_msg=f"{k} not linear in the upstream gradient")
    inv = a["radii"] == 0
    assert not a["dL_dsh"][inv].any() and not a["dL_dmeans3D"][inv].any()
    return D


def test_c3_3m_properties():
    D = _properties(3_000_000, W, H)
    assert 12_000_000 < D < 15_000_000      # SURVEY 8d calibration: D/P ~ 4.5 at 1920x1080


def test_4k_frame_properties():
    """3840x2160 (the C5 resolution) on one GPU with 1.5 M Gaussians: ragged 135-row tile grid, D/P ~ 11."""
    D = _properties(1_500_000, 3840, 2160, check_linearity=False)
    assert D > 10 * 1_500_000 * 0.8


def test_c5_scene_6m_at_4k_properties():
    """BASELINE config 5's scene in full on one GPU: 6 M Gaussians at 3840x2160 (D ~ 67 M): the single-GPU half of C5 (the 8-GPU
    half shards frames, one per GPU, of exactly this workload)."""
    D = _properties(6_000_000, 3840, 2160, check_linearity=False)
    assert 60_000_000 < D < 75_000_000      # SURVEY 8d calibration: D/P ~ 11.2 at 3840x2160


def test_20m_gaussians_properties():
    """Seven times the C3 scene (20 M Gaussians, D ~ 89 M, 37 GB of state): byte offsets beyond 2^32 in every buffer, 19 532 blocks in
    the column pass, 43 642 in the row pass -- same list / image / determinism properties."""
    D = _properties(20_000_000, W, H, check_linearity=False)
    assert 80_000_000 < D < 100_000_000


@pytest.mark.parametrize("tile", [(32, 16), (8, 8)])
def test_tile_shapes_at_full_resolution_against_oracle(tile):
    """BASELINE config 5's tile-size sweep at full 1920x1080 resolution (C2's 500 k Gaussians, so that the CPU oracle finishes in
    seconds): the two extreme shapes bin bit-exactly like the oracle run with the same BLOCK_X x BLOCK_Y and pass the strict bar."""
    from tests.gpu_util import assert_strict_parity, forced_f64_reference, run_hip, run_hip_raw, run_oracle
    P = 500_000
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=0)
    bg = np.zeros(3, np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=1)
    fwd, _ = run_oracle(g, cam, bg, 3, tile=tile)
    raw = run_hip_raw(g, cam, bg, 3, tile=tile)
    assert raw["D"] == fwd["num_rendered"]
    np.testing.assert_array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    out = run_hip(g, cam, bg, 3, dc, da, tile=tile)
    _, fwd64, bwd64 = forced_f64_reference(g, cam, bg, 3, dc, da, tile=tile)
    assert_strict_parity(out, fwd64, bwd64, tag=f"tile {tile} ")


def test_more_than_65536_tiles_against_oracle():
    """3840x2160 with 8x8 tiles = 129 600 tiles (17 key bits): the tile partition runs four radix passes there instead of two;
    list, ranges and images against the oracle with the same tile."""
    from tests.gpu_util import assert_close_frac, run_hip_raw, run_oracle
    P, W4, H4, tile = 150_000, 3840, 2160, (8, 8)
    cam = synthetic_camera(W4, H4)
    g = synthetic_gaussians(P, W4, H4, seed=5)
    bg = np.zeros(3, np.float32)
    fwd, _ = run_oracle(g, cam, bg, 3, tile=tile)
    raw = run_hip_raw(g, cam, bg, 3, tile=tile)
    assert raw["D"] == fwd["num_rendered"] and fwd["ranges"].shape[0] == 480 * 270
    np.testing.assert_array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    assert_close_frac(raw["color"], fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, "4K 8x8 color")
    # the dispatch order over 16 chunks of 8192 tiles: a permutation, length classes descending, index order inside a class
    order = raw["bin"]["tile_order"].view(np.uint32).astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(480 * 270))
    ln = (fwd["ranges"][:, 1].astype(np.int64) - fwd["ranges"][:, 0])[order]
    cls = np.where(ln > 0, np.minimum(15, np.maximum(0, 20 - np.floor(np.log2(np.maximum(ln, 1))).astype(np.int64))), 15)
    assert (np.diff(cls) >= 0).all() and all((np.diff(order[cls == c]) > 0).all() for c in np.unique(cls))
