"""GPU (-m gpu): the HIP path, called through the drop-in package -> C-ABI, against the CPU oracle.

Bars (BASELINE.json north_star): integer / index outputs bit-exact (radii, tiles_touched, depth keys, sorted
(tile, depth, id) lists, tile ranges, contributor counts up to threshold flips); float outputs within 1e-4."""
import math
import os

import numpy as np
import pytest
import torch

from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run through gpurun)"
    from streetunveiler_amd import _lib
    _lib.load()  # fail loudly if the HIP library was not built


def _scene(P, W, H, seed, lo, hi, cam_index=None):
    cam = synthetic_camera(W, H, index=cam_index)
    g = synthetic_gaussians(P, W, H, seed=seed, scale_lo=lo, scale_hi=hi)
    return cam, g


def _check_binning(raw, fwd):
    from tests.gpu_util import np as _np  # noqa: F401
    P = fwd["radii"].shape[0]
    assert raw["D"] == fwd["num_rendered"]
    np.testing.assert_array_equal(raw["radii"], fwd["radii"])
    np.testing.assert_array_equal(raw["geom"]["tiles_touched"].view(np.uint32), fwd["tiles_touched"])
    vis = fwd["radii"] > 0
    keys = raw["geom"]["depth_keys"].view(np.uint32)
    np.testing.assert_array_equal(keys[vis], fwd["depths"][vis].view(np.uint32))
    assert (keys[~vis] == 0xFFFFFFFF).all()
    # packed records: transMat, centre, normal, rgb (K1 runs the oracle's op order -> tight)
    rec = raw["geom"]["splats"]
    np.testing.assert_array_equal(rec[vis][:, 0:9], fwd["transMat"][vis])
    np.testing.assert_array_equal(rec[vis][:, 9:11], fwd["means2D"][vis])
    np.testing.assert_array_equal(rec[vis][:, 12:15], fwd["normal_opacity"][vis][:, :3])
    np.testing.assert_allclose(rec[vis][:, 15:18], fwd["rgb"][vis], atol=2e-6)
    # the sorted duplicate list == the reference's 64-bit (tile<<32 | depth) stable sort, bit for bit
    pl = raw["bin"]["point_list"].view(np.uint32)
    np.testing.assert_array_equal(pl, fwd["point_list"])
    key64 = (raw["bin"]["tile_keys"].view(np.uint32).astype(np.uint64) << np.uint64(32)) | keys[pl].astype(np.uint64)
    np.testing.assert_array_equal(key64, fwd["keys"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    # dispatch order of the blend waves: a permutation of the tiles, power-of-two length classes descending, index order inside
    order = raw["bin"]["tile_order"].view(np.uint32).astype(np.int64)
    assert sorted(order.tolist()) == list(range(fwd["ranges"].shape[0]))
    ln = (fwd["ranges"][:, 1].astype(np.int64) - fwd["ranges"][:, 0])[order]
    cls = np.where(ln > 0, np.minimum(15, np.maximum(0, 20 - np.floor(np.log2(np.maximum(ln, 1))).astype(np.int64))), 15)
    assert (np.diff(cls) >= 0).all() and all((np.diff(order[cls == c]) > 0).all() for c in np.unique(cls))


def _check_images(out, fwd, tag):
    from tests.gpu_util import assert_close_frac
    assert_close_frac(out["color"], fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, tag + " color")
    # allmap: depth-like channels are O(10), use rtol too
    from tests.gpu_util import check_allmap
    check_allmap(out["allmap"], fwd["allmap"], tag)


def _check_grads(out, bwd, names, tag, rel=2e-3):
    from tests.gpu_util import assert_grads_close
    for k in names:
        assert out[k] is not None, k
        assert_grads_close(out[k], bwd[k], rel, f"{tag} {k}")


def test_golden_small_scene(golden_dir):
    """The committed oracle fixture (64 Gaussians, 32x32, SH3): every stage."""
    from tests.gpu_util import run_hip, run_hip_raw, run_oracle
    from streetunveiler_amd.camera import SimpleCamera
    z = np.load(os.path.join(golden_dir, "oracle_small.npz"))
    g = {k: torch.tensor(z["in_" + k]) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
    base = synthetic_camera(32, 32, index=2)
    cam = SimpleCamera(32, 32, base.FoVx, base.FoVy, torch.tensor(z["in_view"]), torch.tensor(z["in_proj"]), torch.tensor(z["in_campos"]))
    dc, da = torch.tensor(z["in_dL_dcolor"]), torch.tensor(z["in_dL_dallmap"])
    raw = run_hip_raw(g, cam, z["in_bg"], 3)
    fwd = {k[4:]: z[k] for k in z.files if k.startswith("fwd_")}
    fwd["num_rendered"] = int(z["fwd_num_rendered"])
    _check_binning(raw, fwd)
    np.testing.assert_array_equal(raw["img"]["n_contrib"].view(np.uint32), fwd["n_contrib"])
    np.testing.assert_allclose(raw["img"]["final_T"], fwd["final_T"], rtol=1e-4, atol=1e-5)
    out = run_hip(g, cam, z["in_bg"], 3, dc, da)
    np.testing.assert_allclose(out["color"], fwd["color"], atol=1e-4)
    np.testing.assert_allclose(out["allmap"], fwd["allmap"], rtol=1e-4, atol=1e-4)
    bwd = {k[4:]: z[k] for k in z.files if k.startswith("bwd_")}
    _check_grads(out, bwd, ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"], "golden")


@pytest.mark.parametrize("P,W,H,deg,lo,hi,cam_index", [
    (20000, 320, 200, 3, 2e-3, 3e-2, 2),      # ragged height (200 = 12.5 tiles), yawed camera
    (5000, 250, 130, 1, 5e-3, 1e-1, None),    # big splats, many tiles per Gaussian, ragged both ways
    (50000, 256, 256, 0, 5e-4, 5e-3, None),   # the C1-style scene (radius floor, thin splats -> low-pass branch)
])
def test_forward_backward_vs_oracle(P, W, H, deg, lo, hi, cam_index):
    cam, g = _scene(P, W, H, P, lo, hi, cam_index)
    _full_check(cam, g, deg, np.array([0.3, 0.1, 0.7], np.float32), synthetic_upstream_grads(W, H, seed=P), f"P{P}")


def _full_check(cam, g, deg, bg, upstream, tag, **budgets):
    """Every stage against the float32 oracle, then the free-running float64 reference."""
    from tests.gpu_util import run_hip, run_hip_raw, run_oracle
    P = g["means3D"].shape[0]
    dc, da = upstream
    fwd, bwd = run_oracle(g, cam, bg, deg, dc, da)
    assert fwd["num_rendered"] > P // 2
    raw = run_hip_raw(g, cam, bg, deg, decisions=True)
    _check_binning(raw, fwd)
    # the two words the emission scan leaves for the forward blend's choice of mapping: D and the Gaussians with at least one tile
    assert raw["geom"]["frame_counts"].view(np.uint32).tolist() == [fwd["num_rendered"], int((fwd["tiles_touched"] > 0).sum())]
    nc = raw["img"]["n_contrib"].view(np.uint32)
    assert (nc != fwd["n_contrib"]).mean() < 1e-3   # contributor counts: equal up to rare threshold flips
    out = run_hip(g, cam, bg, deg, dc, da)
    _check_images(out, fwd, tag)
    _check_grads(out, bwd, ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"], tag)
    # invisible Gaussians get exactly zero gradient
    inv = fwd["radii"] == 0
    for k in ["dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D", "dL_dopacity"]:
        assert not np.asarray(out[k])[inv].any(), k
    # ... and against the free-running float64 reference (its own decisions): 1e-4 and identical stop / median positions at EVERY
    # robust pixel, strict rows on every robust Gaussian, the non-robust remainder counted (tests/gpu_util.py)
    from tests.gpu_util import assert_free_parity, free_f64_reference
    xfwd, xbwd, margins = free_f64_reference(g, cam, bg, deg, dc, da, base=fwd, kernel_decisions=raw["decisions"])
    assert_free_parity(out, nc, xfwd, xbwd, margins, tag=tag + " ", scene=(g, cam), **budgets)


@pytest.mark.parametrize("seed,P,W,H,deg,lo,hi,spread", [
    (1, 30000, 400, 240, 3, 2e-3, 3e-2, 10.0),     # any rotation, camera ~10 units from the origin, FoVx unrelated to FoVy
    (2, 12000, 250, 330, 2, 5e-3, 1e-1, 3.0),      # portrait frame, big splats
    (3, 60000, 320, 320, 3, 5e-4, 5e-3, 60.0),     # thin splats, the camera (and the whole scene) ~60 units out
])
def test_general_camera_poses_vs_oracle(seed, P, W, H, deg, lo, hi, spread):
    """Cameras in GENERAL position (streetunveiler_amd.synthetic.posed_scene): the benchmark cameras of SURVEY 8d sit at the origin and
    only yaw, so `campos`, the translation row of the view matrix and most of its rotation block are zeros and ones in every other test
    -- a K1 / K8 term reading the wrong one of them would pass them all.  Same checks as above, bit-exact integers included."""
    from streetunveiler_amd.synthetic import posed_scene
    cam, g = posed_scene(P, W, H, seed=seed, scale_lo=lo, scale_hi=hi, spread=spread)
    assert float(cam.camera_center.abs().max()) > 0.5 and abs(cam.FoVx - cam.FoVy) > 1e-3
    if seed == 2:   # quaternions "as given" (Appendix A.2): not unit length -- the operator does not normalise them, the model's getter does
        g["rotations"] = (g["rotations"] * (torch.rand(P, 1, generator=torch.Generator().manual_seed(seed)) + 0.6)).contiguous()
    _full_check(cam, g, deg, np.array([0.2, 0.4, 0.1], np.float32), synthetic_upstream_grads(W, H, seed=seed), f"pose{seed}")


def test_cloned_gaussians_and_depth_ties():
    """Densification clones Gaussians in place [REF scene/gaussian_model.py densify_and_clone]: identical means, hence identical depth keys,
    in every tile they touch -- the list order among them is the stable sort's (ascending index), and the blend is order-dependent.  Here
    every position is shared by 8 Gaussians with their own scales / rotations / colours, and a third of the scene sits on four depth
    planes (thousands of equal keys per tile): lists bit-exact against the oracle's 64-bit stable sort, images and gradients as usual."""
    P, W, H = 24000, 320, 200
    cam, g = _scene(P, W, H, 77, 3e-3, 4e-2, 4)
    base = g["means3D"][: P // 8].clone()
    g["means3D"] = base.repeat(8, 1).contiguous()                       # Gaussian i and i + k P/8 share a position
    third = P // 3
    planes = torch.tensor([2.0, 5.0, 11.0, 23.0])[torch.arange(third) % 4]
    world_z_axis = cam.world_view_transform[:3, 2]                      # view depth = p . column 2 (camera at the origin)
    p = g["means3D"][:third]
    depth = p @ world_z_axis
    g["means3D"][:third] = p * (planes / depth)[:, None]                 # along the ray: same pixel, new depth ...
    g["scales"][:third] = g["scales"][:third] * (planes / depth)[:, None]   # ... and the same footprint
    from tests.gpu_util import run_oracle
    fwd, _ = run_oracle(g, cam, np.zeros(3, np.float32), 3)
    keys = fwd["depths"][fwd["radii"] > 0].view(np.uint32)
    assert np.unique(keys).size < 0.5 * keys.size, "the scene was meant to be full of equal depth keys"
    # (eight co-located Gaussians per position make every pixel's list eight times as dense in near-threshold decisions: 2.8 % of the
    # pixels are non-robust here against ~1 % on the plain small scenes -- a classification, not an error; the robust ones hold the usual bars)
    _full_check(cam, g, 3, np.array([0.1, 0.5, 0.2], np.float32), synthetic_upstream_grads(W, H, seed=77), "ties", pixel_budget=5e-2, gaussian_budget=0.6)


def test_colors_precomp_and_transmat_precomp():
    from tests.gpu_util import run_hip, run_oracle
    P, W, H = 3000, 200, 120
    cam, g = _scene(P, W, H, 11, 5e-3, 6e-2, 5)
    bg = np.array([0.0, 1.0, 0.0], np.float32)   # render_semantic-style one-hot background
    dc, da = synthetic_upstream_grads(W, H, seed=3)
    colors = np.random.default_rng(0).random((P, 3)).astype(np.float32)
    fwd, bwd = run_oracle(g, cam, bg, 0, dc, da, colors=colors)
    out = run_hip(g, cam, bg, 0, dc, da, colors=colors)
    np.testing.assert_array_equal(out["radii"], fwd["radii"])
    _check_images(out, fwd, "colors_precomp")
    _check_grads(out, bwd, ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcolors", "dL_dmeans2D"], "colors_precomp")
    # precomputed transMat (the reference's cov3D_precomp slot)
    Tpre = fwd["transMat"].copy()
    Tpre[fwd["radii"] == 0] = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    fwd2, bwd2 = run_oracle(g, cam, bg, 2, dc, da, Tpre=Tpre)
    out2 = run_hip(g, cam, bg, 2, dc, da, Tpre=Tpre)
    np.testing.assert_array_equal(out2["radii"], fwd2["radii"])
    _check_images(out2, fwd2, "transMat_precomp")
    _check_grads(out2, bwd2, ["dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dtransMat", "dL_dmeans2D"], "transMat_precomp")


def test_empty_and_all_culled_inputs():
    from tests.gpu_util import run_hip
    W, H = 64, 48
    cam = synthetic_camera(W, H)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    dc, da = synthetic_upstream_grads(W, H)
    # P == 0
    g0 = {k: v[:0] for k, v in synthetic_gaussians(4, W, H).items()}
    out = run_hip(g0, cam, bg, 3, dc, da)
    np.testing.assert_allclose(out["color"], np.broadcast_to(bg[:, None, None], (3, H, W)))
    assert not out["allmap"].any() and out["radii"].shape == (0,)
    # everything behind the camera: D == 0
    g = synthetic_gaussians(100, W, H)
    g["means3D"][:, 2] *= -1
    out = run_hip(g, cam, bg, 3, dc, da)
    assert not out["radii"].any()
    np.testing.assert_allclose(out["color"], np.broadcast_to(bg[:, None, None], (3, H, W)))
    for k in ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
        assert not out[k].any(), k


def test_mark_visible_and_debug_mode():
    from diff_surfel_rasterization import GaussianRasterizer
    from oracle import surfel_oracle as so
    from tests.gpu_util import DEV, run_hip, settings_for
    W, H = 96, 64
    cam, g = _scene(2000, W, H, 5, 2e-3, 3e-2, 1)
    g["means3D"][::3, 2] *= -1
    vis = GaussianRasterizer(settings_for(cam, [0, 0, 0], 0)).markVisible(g["means3D"].to(DEV))
    assert vis.dtype == torch.bool
    np.testing.assert_array_equal(vis.cpu().numpy(), so.mark_visible(g["means3D"].numpy(), cam.world_view_transform.numpy()))
    # ... through a camera with its own centre and a full rotation, standing inside the cloud (60 % of the Gaussians behind it); and the
    # independent statement of the rule: view depth > 0.2 with the depth computed in float64
    from streetunveiler_amd.synthetic import posed_scene
    camp, gp = posed_scene(5000, W, H, seed=9, spread=30.0, behind_fraction=0.6)
    visp = GaussianRasterizer(settings_for(camp, [0, 0, 0], 0)).markVisible(gp["means3D"].to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(visp, so.mark_visible(gp["means3D"].numpy(), camp.world_view_transform.numpy()))
    z64 = gp["means3D"].double().numpy() @ camp.world_view_transform.double().numpy()[:3, 2] + camp.world_view_transform.double().numpy()[3, 2]
    sure = np.abs(z64 - 0.2) > 1e-4
    assert 0.3 < visp.mean() < 0.5 and np.array_equal(visp[sure], (z64 > 0.2)[sure])
    # debug=True: sync-and-check after every kernel, same numbers
    a = run_hip(g, cam, [0, 0, 0], 2, debug=False)
    b = run_hip(g, cam, [0, 0, 0], 2, debug=True)
    np.testing.assert_array_equal(a["color"], b["color"])


def test_quadrant_culling_is_exact():
    """The per-quadrant culling only drops entries the per-pixel test would skip: images, per-pixel state and
    gradients with culling ON must equal culling OFF (images/state bit for bit; gradients up to the
    order of the 4-wave LDS combine)."""
    from tests.gpu_util import run_hip, run_hip_raw
    for (P, W, H, lo, hi, idx) in [(30000, 384, 216, 5e-4, 5e-3, None), (8000, 200, 150, 5e-3, 8e-2, 6), (3000, 160, 96, 2e-2, 3e-1, 1)]:
        cam, g = _scene(P, W, H, P + 1, lo, hi, idx)
        g["opacities"][::7] = 0.003   # below 1/255: can never contribute
        g["opacities"][::11] = 1.0
        dc, da = synthetic_upstream_grads(W, H, seed=P)
        res = {}
        for cull in (1, 0):   # per call: SrFrame.flags & SR_FLAG_NO_QUADRANT_CULL
            res[cull] = (run_hip_raw(g, cam, [0.2, 0.4, 0.6], 3, quadrant_cull=bool(cull)),
                         run_hip(g, cam, [0.2, 0.4, 0.6], 3, dc, da, quadrant_cull=bool(cull)))
        (raw1, out1), (raw0, out0) = res[1], res[0]
        np.testing.assert_array_equal(raw1["color"], raw0["color"])
        np.testing.assert_array_equal(raw1["allmap"], raw0["allmap"])
        np.testing.assert_array_equal(raw1["img"]["n_contrib"], raw0["img"]["n_contrib"])
        np.testing.assert_array_equal(raw1["img"]["final_T"], raw0["img"]["final_T"])
        for k in ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
            scale = np.abs(out0[k]).max() + 1e-20
            assert np.abs(out1[k] - out0[k]).max() <= 2e-5 * scale, k


def test_row_mapped_forward_is_bit_identical():
    """The two forward blend kernels of the 16x16 tile (the device picks one per frame; SR_FLAG_ROW_MAPPED_FORWARD /
    SR_FLAG_QUADRANT_MAPPED_FORWARD force one): the four 16-lane rows of a wave walk the lists of four 4x4 cells instead of one entry on 64 lanes --
    the same per-pixel sequence of operations, so images, per-pixel state and (through the exact hit masks the backward visits)
    every gradient equal the default kernel's bit for bit; shapes it does not exist for are refused by name."""
    from streetunveiler_amd import _lib
    from tests.gpu_util import run_hip, run_hip_raw
    for (P, W, H, lo, hi, idx) in [(30000, 384, 216, 5e-4, 5e-3, None), (8000, 200, 150, 5e-3, 8e-2, 6), (3000, 161, 97, 2e-2, 3e-1, 1)]:
        cam, g = _scene(P, W, H, P + 2, lo, hi, idx)
        g["opacities"][::11] = 1.0
        dc, da = synthetic_upstream_grads(W, H, seed=P)
        raw0, raw1, raw2 = (run_hip_raw(g, cam, [0.2, 0.4, 0.6], 3, row_mapped=r) for r in (False, True, None))   # quadrants, rows, the device's pick
        out0, out1, out2 = (run_hip(g, cam, [0.2, 0.4, 0.6], 3, dc, da, row_mapped=r) for r in (False, True, None))
        np.testing.assert_array_equal(raw2["color"], raw0["color"]); np.testing.assert_array_equal(raw2["img"]["n_contrib"], raw0["img"]["n_contrib"])
        np.testing.assert_array_equal(out2["dL_dmeans3D"], out0["dL_dmeans3D"])
        for k in ("color", "allmap"):
            np.testing.assert_array_equal(raw1[k], raw0[k])
        np.testing.assert_array_equal(raw1["img"]["n_contrib"], raw0["img"]["n_contrib"])
        np.testing.assert_array_equal(raw1["img"]["final_T"], raw0["img"]["final_T"])
        for k in ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]:
            np.testing.assert_array_equal(out1[k], out0[k], err_msg=k)
    with pytest.raises(_lib.SurfelRasterError, match="ROW_MAPPED"):
        run_hip_raw(g, cam, [0.2, 0.4, 0.6], 3, tile=(8, 8), row_mapped=True)


def test_lds_atomic_returns_in_lane_order():
    """The rank phase of the sort / partition kernels is ONE LDS atomic per item: it relies on ds_add_rtn_u32 handing its return values
    to the lanes of a wave instruction that hit the same address in ascending lane order (not documented for gfx950).  Checked here
    on the device for digit alphabets from 1 (every lane the same address) to 1000, 2 M lane-items each."""
    from streetunveiler_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n = 256 * 8 * 1024
    for bins in (1, 2, 3, 7, 16, 68, 120, 256, 1000):
        g = torch.Generator().manual_seed(bins)
        digits = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        ranks = torch.empty(n, dtype=torch.int32, device=dev)
        assert lib.sr_debug_lds_atomic_ranks(digits.data_ptr(), ranks.data_ptr(), n, bins, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        d = (digits.cpu().numpy().view(np.uint32) % np.uint32(bins)).reshape(-1, 8, 4, 64)      # block, step, wave, lane
        r = ranks.cpu().numpy().reshape(-1, 8, 4, 64)
        # expected: the number of earlier (step, lane) items of the same wave with the same digit
        d2 = d.transpose(0, 2, 1, 3).reshape(-1, 512)    # per wave: items in issue order (step-major, lane-minor)
        r2 = r.transpose(0, 2, 1, 3).reshape(-1, 512)
        order = np.argsort(d2, axis=1, kind="stable")
        ds = np.take_along_axis(d2, order, axis=1)
        first = np.concatenate([np.ones((ds.shape[0], 1), bool), ds[:, 1:] != ds[:, :-1]], axis=1)
        pos = np.arange(512)[None, :]
        start = np.maximum.accumulate(np.where(first, pos, 0), axis=1)
        expect = np.empty_like(r2)
        np.put_along_axis(expect, order, (pos - start).astype(r2.dtype), axis=1)
        np.testing.assert_array_equal(r2, expect, err_msg=f"bins={bins}")


def test_radix_sort_stability_and_edges():
    """The hand-written LSD radix sort behind K2/K4: stable, correct at ragged sizes and for every bit count."""
    import ctypes as C
    from streetunveiler_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    cases = [(1, 32), (63, 5), (64, 13), (65, 8), (2047, 7), (2048, 16), (2049, 9), (100_003, 13), (1_000_001, 32), (777_777, 3)]
    # digits shared by most lanes of a row (the peeled path of common.h take_run_slot / wave_count_digit): one key everywhere, one bit,
    # and depth keys -- the bits of floats between 0.5 and 50, whose top byte takes four values
    cases += [(70_001, 0), (300_001, 1), (2_100_000, -1)]
    # the one-sweep passes (32 bits from 2^18 items up: look-back over the blocks' published digit counts): the threshold itself, one key
    # everywhere (every block's whole count in one digit), and a size whose last block is ragged
    cases += [(1 << 18, 32), ((1 << 18) + 1, 0), (4_194_304 + 77, -1)]
    for n, bits in cases:
        if bits == -1:
            keys, bits = rng.uniform(0.5, 50.0, size=n).astype(np.float32).view(np.uint32), 32
        elif bits == 0:
            keys, bits = np.full(n, 0xDEADBEEF, np.uint32), 32
        else:
            keys = rng.integers(0, 2 ** bits, size=n, dtype=np.uint64).astype(np.uint32)
        for vals in (None, rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)):
            k = torch.from_numpy(keys.view(np.int32)).to(dev)
            v = None if vals is None else torch.from_numpy(vals.view(np.int32)).to(dev)
            ko, vo = torch.empty_like(k), torch.empty_like(k)
            tmp = torch.empty(lib.sr_debug_radix_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
            # the LDS-atomic ranking and its match-any fallback, classic three-launch passes and one-sweep passes (32 bits, >= 2^18 items)
            for flags in (0, _lib.SR_FLAG_BALLOT_RANKING, _lib.SR_FLAG_ONE_SWEEP_SORT, _lib.SR_FLAG_ONE_SWEEP_SORT | _lib.SR_FLAG_BALLOT_RANKING):
                if (flags & _lib.SR_FLAG_ONE_SWEEP_SORT) and not (bits == 32 and n >= (1 << 18)):
                    continue
                rc = lib.sr_debug_radix_sort(k.data_ptr(), None if v is None else v.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, bits,
                                             tmp.data_ptr(), tmp.numel(), flags, C.c_void_p(torch.cuda.current_stream().cuda_stream))
                _lib.check(rc, "sr_debug_radix_sort")
                torch.cuda.synchronize()
                order = np.argsort(keys, kind="stable")
                np.testing.assert_array_equal(ko.cpu().numpy().view(np.uint32), keys[order])
                expect_v = order.astype(np.uint32) if vals is None else vals[order]
                np.testing.assert_array_equal(vo.cpu().numpy().view(np.uint32), expect_v)


def test_rank_self_check_and_forced_ballot_fallback():
    """The run-time guard of the LDS-atomic lane-order assumption: the library's own self-check (run on first use per device) selects a
    ranking, and with the match-any fallback FORCED (SrFrame.flags & SR_FLAG_BALLOT_RANKING) the depth order, the duplicate list, the
    tile ranges and the images are bit-identical to the oracle's / the default path's -- for the 16x16 tile and for a shape whose
    partition digits are wider (8x8 at 640 px: 80 columns)."""
    from streetunveiler_amd import _lib
    from tests.gpu_util import DEV, run_hip_raw, run_oracle, settings_for
    from diff_surfel_rasterization import _C
    lib = _lib.load()
    mode = lib.sr_rank_mode(C_void(torch.cuda.current_stream().cuda_stream))
    assert mode in (1, 2), f"sr_rank_mode: {mode} ({lib.sr_last_error()})"
    assert mode == 1, "gfx950 returns LDS atomic results in lane order: the fast ranking should have been selected"
    for (P, W, H, tile) in [(40000, 640, 360, None), (15000, 640, 200, (8, 8))]:
        cam, g = _scene(P, W, H, 77, 1e-3, 2e-2)
        bg = np.array([0.1, 0.0, 0.4], np.float32)
        fwd, _ = run_oracle(g, cam, bg, 2, tile=tile or (16, 16))
        s = settings_for(cam, bg, 2)
        e = torch.empty(0, device=DEV)
        d = lambda k: g[k].to(DEV)
        outs = []
        for ballot in (True, False):
            D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
                s.bg, d("means3D"), e, d("opacities"), d("scales"), d("rotations"), 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                s.image_height, s.image_width, d("shs"), 2, s.campos, False, False, tile=tile, ballot_ranking=ballot)
            torch.cuda.synchronize()
            bv = _C.binning_view(binning, P, D, W, H, tile or (16, 16))
            gv = _C.geom_view(geom, P)
            outs.append((D, gv["sorted_gid"].cpu().numpy().copy(), bv["point_list"].cpu().numpy().copy(), bv["ranges"].cpu().numpy().copy(), color.cpu().numpy()))
        for D, sorted_gid, pl, ranges, color in outs:
            assert D == fwd["num_rendered"]
            np.testing.assert_array_equal(pl.view(np.uint32), fwd["point_list"])
            np.testing.assert_array_equal(ranges.view(np.uint32), fwd["ranges"])
        V = int((fwd["radii"] > 0).sum())   # (the depth sort compacts: only the first V entries of its outputs are written)
        np.testing.assert_array_equal(outs[0][1][:V], outs[1][1][:V])
        order = np.argsort(fwd["depths"].view(np.uint32)[fwd["radii"] > 0], kind="stable")
        np.testing.assert_array_equal(outs[0][1][:V], np.flatnonzero(fwd["radii"] > 0)[order].astype(outs[0][1].dtype))
        np.testing.assert_array_equal(outs[0][4], outs[1][4])


def C_void(x):
    import ctypes
    return ctypes.c_void_p(x)


def test_sh_layout_fallbacks_and_scale_modifier():
    """Paths the bench never takes: SH tensors with M != 16 or a 4-byte-aligned base (no LDS row staging), SH degree 2,
    and scale_modifier != 1 (forward honours it; the backward keeps upstream's modifier-free scale gradient, A.6)."""
    import math
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import surfel_oracle as so
    from tests.gpu_util import DEV, assert_close_frac, assert_grads_close
    P, W, H = 3000, 160, 96
    cam, g = _scene(P, W, H, 17, 4e-3, 5e-2, 3)
    bg = np.array([0.1, 0.0, 0.2], np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=9)
    for (M, deg, misalign, modifier) in [(4, 1, False, 1.0), (9, 2, False, 1.0), (16, 2, True, 1.0), (16, 3, False, 0.7)]:
        shs_cpu = g["shs"][:, :M].contiguous()
        fwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), shs=shs_cpu.numpy(),
                                   viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                                   campos=cam.camera_center.numpy(), bg=bg, image_width=W, image_height=H, sh_degree=deg, scale_modifier=modifier,
                                   tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
        bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy())
        if misalign:   # a view that starts 4 bytes into its storage: contiguous but not 16-byte aligned
            buf = torch.zeros(P * M * 3 + 1, device=DEV)
            shs = buf[1:].view(P, M, 3)
            shs.copy_(shs_cpu.to(DEV))
            assert shs.data_ptr() % 16 != 0
            shs.requires_grad_()
        else:
            shs = shs_cpu.to(DEV).requires_grad_()
        t = {k: g[k].to(DEV).requires_grad_() for k in ["means3D", "opacities", "scales", "rotations"]}
        s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.tensor(bg).to(DEV), modifier,
                                          cam.world_view_transform.to(DEV), cam.full_proj_transform.to(DEV), deg, cam.camera_center.to(DEV), False, False)
        m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
        color, radii, allmap = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, shs=shs, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([color, allmap], [dc.to(DEV), da.to(DEV)])
        tag = f"M{M} deg{deg} misalign{misalign} mod{modifier}"
        np.testing.assert_array_equal(radii.cpu().numpy(), fwd["radii"], err_msg=tag)
        assert_close_frac(color.detach().cpu().numpy(), fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, tag)
        for name, got in [("dL_dmeans3D", t["means3D"].grad), ("dL_dopacity", t["opacities"].grad), ("dL_dscales", t["scales"].grad),
                          ("dL_drotations", t["rotations"].grad), ("dL_dsh", shs.grad), ("dL_dmeans2D", m2d.grad)]:
            assert_grads_close(got.cpu().numpy(), bwd[name], 2e-3, tag + " " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("use_tpre", [False, True])
def test_six_colour_channels_equal_two_three_channel_passes(use_tpre):
    """one 6-channel pass == the reference's two 3-channel passes over the same geometry
    (/root/reference/gaussian_renderer/__init__.py:417-444).  Oracle pass A carries colours 0..2 and the allmap
    gradient, pass B colours 3..5 with a zero allmap gradient; parameter gradients add, colour gradients concatenate."""
    from tests.gpu_util import assert_close_frac, run_hip, run_oracle
    P, W, H = 3000, 208, 120
    cam, g = _scene(P, W, H, 21, 5e-3, 6e-2, 4)
    rng = np.random.default_rng(5)
    colors = rng.random((P, 6)).astype(np.float32)
    bg = np.array([0.0, 0.25, 0.0, 0.0, 1.0, 0.5], np.float32)
    dcA, da = synthetic_upstream_grads(W, H, seed=6)
    dcB, _ = synthetic_upstream_grads(W, H, seed=7)
    dc = torch.cat([dcA, dcB], 0)
    Tpre = None
    if use_tpre:
        f0, _ = run_oracle(g, cam, bg[:3], 0, colors=colors[:, :3].copy())
        Tpre = f0["transMat"].copy()
        Tpre[f0["radii"] == 0] = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    fA, bA = run_oracle(g, cam, bg[:3], 0, dcA, da, colors=colors[:, :3].copy(), Tpre=Tpre)
    fB, bB = run_oracle(g, cam, bg[3:], 0, dcB, torch.zeros_like(da), colors=colors[:, 3:].copy(), Tpre=Tpre)
    out = run_hip(g, cam, bg, 0, dc, da, colors=colors, Tpre=Tpre)
    assert out["color"].shape == (6, H, W)
    np.testing.assert_array_equal(out["radii"], fA["radii"])
    fwd = dict(color=np.concatenate([fA["color"], fB["color"]], 0), allmap=fA["allmap"])
    _check_images(out, fwd, "six channels")
    names = ["dL_dmeans3D", "dL_dopacity", "dL_dmeans2D"] + (["dL_dtransMat"] if use_tpre else ["dL_dscales", "dL_drotations"])
    bwd = {k: bA[k] + bB[k] for k in names}
    bwd["dL_dcolors"] = np.concatenate([bA["dL_dcolors"], bB["dL_dcolors"]], 1)
    _check_grads(out, bwd, names + ["dL_dcolors"], "six channels")
    # the first three channels and the allmap are bit-identical to a plain 3-channel call on the same inputs
    out3 = run_hip(g, cam, bg[:3], 0, colors=colors[:, :3].copy(), Tpre=Tpre)
    np.testing.assert_array_equal(out["color"][:3], out3["color"])
    np.testing.assert_array_equal(out["allmap"], out3["allmap"])


@pytest.mark.parametrize("tile", [(8, 8), (16, 8), (32, 8), (32, 16)])
def test_multi_colour_passes_on_other_tile_shapes(tile):
    """The shared-geometry passes on the other tile shapes of BASELINE config 5's sweep: the 6-channel pass == two 3-channel
    oracle passes run with the same tile; the 9-channel pass (SH colour + six channels) reproduces the 3-channel SH render bit for bit in
    its first three channels and the 6-channel pass in the other six.  (32x16: eight pixels per lane -- the forward runs two 32x8 band
    waves per tile, the backward walks the list once per band and adds the second walk's sums to the first one's records.)"""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import DEV, run_hip, run_oracle, settings_for
    P, W, H = 2500, 208, 120
    cam, g = _scene(P, W, H, 23, 5e-3, 6e-2, 4)
    rng = np.random.default_rng(6)
    colors = rng.random((P, 6)).astype(np.float32)
    bg = np.array([0.0, 0.25, 0.0, 0.0, 1.0, 0.5], np.float32)
    dcA, da = synthetic_upstream_grads(W, H, seed=6)
    dcB, _ = synthetic_upstream_grads(W, H, seed=7)
    fA, bA = run_oracle(g, cam, bg[:3], 0, dcA, da, colors=colors[:, :3].copy(), tile=tile)
    fB, bB = run_oracle(g, cam, bg[3:], 0, dcB, torch.zeros_like(da), colors=colors[:, 3:].copy(), tile=tile)
    out = run_hip(g, cam, bg, 0, torch.cat([dcA, dcB], 0), da, colors=colors, tile=tile)
    assert out["color"].shape == (6, H, W)
    np.testing.assert_array_equal(out["radii"], fA["radii"])
    _check_images(out, dict(color=np.concatenate([fA["color"], fB["color"]], 0), allmap=fA["allmap"]), f"six channels {tile}")
    names = ["dL_dmeans3D", "dL_dopacity", "dL_dmeans2D", "dL_dscales", "dL_drotations"]
    bwd = {k: bA[k] + bB[k] for k in names}
    bwd["dL_dcolors"] = np.concatenate([bA["dL_dcolors"], bB["dL_dcolors"]], 1)
    _check_grads(out, bwd, names + ["dL_dcolors"], f"six channels {tile}")
    # nine channels: SH colour + the six precomputed ones in one pass, against the separate calls of the SAME build and tile
    t = {k: v.to(DEV) for k, v in g.items()}
    geo = dict(means3D=t["means3D"], means2D=torch.zeros(P, 3, device=DEV), opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    bg9 = np.concatenate([[0.1, 0.2, 0.3], bg]).astype(np.float32)
    c9, r9, a9 = GaussianRasterizer(settings_for(cam, bg9, 3), tile=tile)(shs=t["shs"], extra_colors=torch.as_tensor(colors).to(DEV), **geo)
    c3, r3, a3 = GaussianRasterizer(settings_for(cam, bg9[:3], 3), tile=tile)(shs=t["shs"], **geo)
    c6, _, _ = GaussianRasterizer(settings_for(cam, bg, 0), tile=tile)(colors_precomp=torch.as_tensor(colors).to(DEV), **geo)
    assert torch.equal(c9[:3], c3) and torch.equal(c9[3:], c6) and torch.equal(a9, a3) and torch.equal(r9, r3)
    # ... and its backward against the two separate calls' (9 channels = SH colour gradients + the six precomputed ones)
    dc9 = torch.cat([dcA, dcA.flip(0), dcB], 0).to(DEV)
    leaves = lambda: {k: t[k].clone().requires_grad_() for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    def run(fn):
        l = leaves(); ex = torch.as_tensor(colors).to(DEV).requires_grad_()
        m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
        fn(l, ex, m2d).backward()
        return {**{k: v.grad for k, v in l.items()}, "extra": ex.grad, "m2d": m2d.grad}
    gkw = lambda l, m2d: dict(means3D=l["means3D"], means2D=m2d, opacities=l["opacities"], scales=l["scales"], rotations=l["rotations"])
    def nine(l, ex, m2d):
        c, _, a = GaussianRasterizer(settings_for(cam, bg9, 3), tile=tile)(shs=l["shs"], extra_colors=ex, **gkw(l, m2d))
        return (c * dc9).sum() + (a * da.to(DEV)).sum()
    def separate(l, ex, m2d):
        c3_, _, a3_ = GaussianRasterizer(settings_for(cam, bg9[:3], 3), tile=tile)(shs=l["shs"], **gkw(l, m2d))
        c6_, _, _ = GaussianRasterizer(settings_for(cam, bg, 0), tile=tile)(colors_precomp=ex, **gkw(l, m2d))
        return (c3_ * dc9[:3]).sum() + (c6_ * dc9[3:]).sum() + (a3_ * da.to(DEV)).sum()
    g9, gs = run(nine), run(separate)
    from tests.gpu_util import assert_grads_close
    for k in g9:
        assert float(gs[k].abs().max()) > 0, k
        assert_grads_close(g9[k].cpu().numpy(), gs[k].cpu().numpy(), 2e-5, f"nine channels {tile} vs separate calls d{k}", max_bad_frac=1e-4, hard=1e-3)


def test_sh_gradient_expand_matches_backward():
    """Frame-parallel SH gradient: K8 with a deferred SH expansion + sr_sh_gradient_expand == K8's own dL_dsh
    (bit for bit with one view), and the expansion of two views == the sum of their dL_dsh."""
    from diff_surfel_rasterization import _C
    from tests.gpu_util import DEV, settings_for
    P, W, H = 5000, 192, 112
    _, g = _scene(P, W, H, 31, 5e-3, 6e-2, 0)
    dc, da = synthetic_upstream_grads(W, H, seed=9)
    e = torch.empty(0, device=DEV)
    d = {k: v.to(DEV) for k, v in g.items()}
    per_view = []
    for index in (1, 6):
        cam = synthetic_camera(W, H, index=index)
        for deg in ((3, 1) if index == 1 else (3,)):
            s = settings_for(cam, np.zeros(3, np.float32), deg)
            D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
                s.bg, d["means3D"], e, d["opacities"], d["scales"], d["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                s.tanfovx, s.tanfovy, H, W, d["shs"], deg, s.campos, False, False)
            args = (s.bg, d["means3D"], radii, e, d["scales"], d["rotations"], 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                    s.tanfovy, dc.to(DEV), da.to(DEV), d["shs"], deg, s.campos, geom, D, binning, img, False)
            full = _C.rasterize_gaussians_backward(*args)
            lean = _C.rasterize_gaussians_backward(*args, defer_sh=True)
            assert lean[5].numel() == 0 and tuple(lean[1].shape) == (P, 3) and full[1].numel() == 0
            for k in (0, 2, 3, 6, 7):   # every other gradient is untouched by the deferral
                assert torch.equal(full[k], lean[k])
            one = _C.sh_gradient_expand(d["means3D"], s.campos, lean[1], 16, deg)
            assert float(full[5].abs().max()) > 0
            assert torch.equal(one, full[5]), f"deg {deg}: max diff {float((one - full[5]).abs().max())}"
            assert not one[(radii == 0)].any()
            if deg == 3:
                per_view.append((s.campos.clone(), lean[1].clone(), full[5].clone()))
    both = _C.sh_gradient_expand(d["means3D"], torch.stack([v[0] for v in per_view]), torch.stack([v[1] for v in per_view]), 16, 3)
    torch.testing.assert_close(both, per_view[0][2] + per_view[1][2], rtol=1e-5, atol=1e-6)
    # ... and with cameras that do NOT share a centre (the yawed batch above does: the origin): three views of synthetic.posed_rig
    from streetunveiler_amd.synthetic import posed_rig
    rig, gp = posed_rig(P, W, H, 3, seed=13, scale_lo=5e-3, scale_hi=6e-2, spread=15.0)
    dp = {k: v.to(DEV) for k, v in gp.items()}
    views = []
    for cam in rig:
        s = settings_for(cam, np.zeros(3, np.float32), 3)
        D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
            s.bg, dp["means3D"], e, dp["opacities"], dp["scales"], dp["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
            s.tanfovx, s.tanfovy, H, W, dp["shs"], 3, s.campos, False, False)
        args = (s.bg, dp["means3D"], radii, e, dp["scales"], dp["rotations"], 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                s.tanfovy, dc.to(DEV), da.to(DEV), dp["shs"], 3, s.campos, geom, D, binning, img, False)
        full = _C.rasterize_gaussians_backward(*args); lean = _C.rasterize_gaussians_backward(*args, defer_sh=True)
        assert float(full[5].abs().max()) > 0 and torch.equal(_C.sh_gradient_expand(dp["means3D"], s.campos, lean[1], 16, 3), full[5])
        views.append((s.campos.clone(), lean[1].clone(), full[5].clone()))
    assert float((views[0][0] - views[1][0]).abs().max()) > 0.1
    three = _C.sh_gradient_expand(dp["means3D"], torch.stack([v[0] for v in views]), torch.stack([v[1] for v in views]), 16, 3)
    torch.testing.assert_close(three, views[0][2] + views[1][2] + views[2][2], rtol=1e-5, atol=1e-6 * float(three.abs().max()))
    swapped = _C.sh_gradient_expand(dp["means3D"], torch.stack([views[1][0], views[0][0], views[2][0]]), torch.stack([v[1] for v in views]), 16, 3)
    assert float((swapped - three).abs().max()) > 1e-3 * float(three.abs().max()), "the expansion does not depend on which camera a view's gradient is paired with"
    # general-layout path (M = 9 rows, degree 2)
    sh9 = d["shs"][:, :9].contiguous()
    nine = _C.sh_gradient_expand(d["means3D"], per_view[0][0], per_view[0][1], 9, 2)
    ref = _C.sh_gradient_expand(d["means3D"], per_view[0][0], per_view[0][1], 16, 2)
    assert torch.equal(nine, ref[:, :9]) and sh9.shape[1] == 9


@pytest.mark.parametrize("rig", ["yawed", "posed"])
def test_factored_sh_exchange_two_ranks_one_gpu(rig):
    """The N > 1 exchange end to end (autograd hook, all-gather, HIP expansion, all-reduce of the rest) with two ranks
    sharing this box's one GPU over gloo; tools/check_factored_exchange.py compares against locally summed gradients.
    `posed`: three ranks whose cameras each have their own centre and orientation (the benchmark's yawed batch shares one centre, the
    origin, so the per-view direction normalize(mean - campos_v) of the expansion is the same for every view there)."""
    import socket, subprocess, sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SURFEL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SURFEL_CHECK_RIG=rig)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2" if rig == "yawed" else "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "check_factored_exchange.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "factored exchange OK" in r.stdout


def test_factored_sh_exchange_over_rccl_with_one_rank():
    """The frame-parallel exchange over the real backend (`nccl` = RCCL) on this one GPU: a one-rank group runs the same calls as an
    N-rank one -- all-gather of the colour gradients on the communication stream, started between K7 and K8, local SH expansion, the flat
    40-B all-reduce -- and must reproduce the local gradients (tools/check_factored_exchange.py, SURFEL_EXCHANGE_SINGLE_RANK=1)."""
    import socket, subprocess, sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SURFEL_EXCHANGE_SINGLE_RANK="1", SURFEL_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_factored_exchange.py")], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "factored exchange OK (world 1, backend nccl)" in r.stdout


def test_k1_rotation_scaling_and_fused_activations_match_reference_fixtures(golden_dir):
    """G7 / G8 on the GPU: K1's splat -> screen transform carries L = R diag(s) of the reference's build_scaling_rotation
    [REF utils/general_utils.py:78-110], and the activations K1 fuses (SR_ACT_*) are the reference GaussianModel's getters
    [REF scene/gaussian_model.py:31-39, 101-123] -- both from fixtures the reference's own python produced (tests/golden/make_golden.py)."""
    import math
    from diff_surfel_rasterization import GaussianRasterizer
    from streetunveiler_amd.camera import make_camera
    from tests.gpu_util import run_hip_raw, settings_for
    z = np.load(os.path.join(golden_dir, "rotation_checkpoint_golden.npz"))
    q = torch.tensor(z["g7_quat"])
    qn = q / torch.sqrt((q * q).sum(1))[:, None]
    n = qn.shape[0]
    W, H = 64, 48
    cam = synthetic_camera(W, H, index=2)
    g = dict(means3D=torch.tensor([[0.1, -0.2, 6.0]]).repeat(n, 1), opacities=torch.full((n, 1), 0.5),
             scales=torch.tensor(z["g7_scale3"][:, :2]) * 0.05, rotations=qn.float().contiguous(), shs=torch.zeros(n, 16, 3))
    raw = run_hip_raw(g, cam, [0, 0, 0], 0)
    rec = raw["geom"]["splats"].reshape(n, 20)
    Tm = rec[:, :9].reshape(n, 3, 3)
    proj = cam.full_proj_transform.numpy().astype(np.float64).reshape(16)
    B = np.zeros((3, 4))
    for k in range(4):
        a0, a1, a3 = proj[4 * k], proj[4 * k + 1], proj[4 * k + 3]
        B[0, k] = 0.5 * W * a0 + 0.5 * (W - 1) * a3; B[1, k] = 0.5 * H * a1 + 0.5 * (H - 1) * a3; B[2, k] = a3
    expect = np.einsum("rk,nkc->nrc", B[:, :3], z["g7_L"].astype(np.float64)[:, :, :2] * 0.05)
    vis = raw["radii"] > 0
    assert vis.sum() > n // 2
    np.testing.assert_allclose(Tm[vis][:, :, :2], expect[vis], rtol=2e-5, atol=2e-5 * np.abs(expect).max())
    # G8: raw checkpoint parameters through the FUSED activations == the reference's activated getters fed to the plain operator
    P = z["g8_raw_xyz"].shape[0]
    cam2 = make_camera(W, H, cam.FoVx, cam.FoVy, t=np.array([0.0, 0.0, 6.0]))
    feats = torch.tensor(z["g8_get_features"])
    common = dict(means3D=torch.tensor(z["g8_raw_xyz"]), shs=feats)
    act = dict(common, opacities=torch.tensor(z["g8_get_opacity"]), scales=torch.tensor(z["g8_get_scaling"]) * 8, rotations=torch.tensor(z["g8_get_rotation"]))
    dev = "cuda:0"
    s = settings_for(cam2, [0.1, 0.2, 0.3], 3)
    plain = GaussianRasterizer(s)(means3D=act["means3D"].to(dev), means2D=torch.zeros(P, 3, device=dev), shs=act["shs"].to(dev),
                                  opacities=act["opacities"].to(dev), scales=act["scales"].to(dev), rotations=act["rotations"].to(dev))
    fused = GaussianRasterizer(s, fused_activations=True)(
        means3D=act["means3D"].to(dev), means2D=torch.zeros(P, 3, device=dev), shs=act["shs"].to(dev), opacities=torch.tensor(z["g8_raw_opacity"]).to(dev),
        scales=(torch.tensor(z["g8_raw_scaling"]) + math.log(8.0)).to(dev), rotations=torch.tensor(z["g8_raw_rotation"]).to(dev))
    assert (plain[1] > 0).sum() >= P // 3
    same = (plain[1] == fused[1]).float().mean().item()
    assert same >= 1 - 1.5 / P, f"radii equal on {same:.3f}"       # (an ulp of exp can cross a ceil)
    if same == 1.0:
        torch.testing.assert_close(fused[0], plain[0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(fused[2], plain[2], rtol=1e-4, atol=1e-4)
    else:   # one radius crossed a ceil(): that splat's rectangle differs by a tile ring -- everything else must still agree
        for a, b, atol in ((fused[0], plain[0], 1e-5), (fused[2], plain[2], 1e-4)):
            bad = ((a - b).abs() > atol + 1e-4 * b.abs()).float().mean().item()
            assert bad < 2e-2, f"fused activations: {bad:.4f} of the elements differ with one radius flipped"


def _bare_env():
    """The environment of a plain `python bench.py` call: no torchrun variables (the test process may itself run under a launcher)."""
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT",
                                                              "TORCHELASTIC_RUN_ID", "SURFEL_EXCHANGE_SINGLE_RANK")}


@pytest.mark.parametrize("extra", [[], ["--frames-per-rank", "2"], ["--exchange", "allreduce"]])
def test_bench_gpus_2_launches_itself_and_exchanges(extra):
    """`python bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's 1-GPU command line): bench.py starts its two
    ranks itself; here they share this box's one GPU over gloo.  The JSON line must report two ranks that both see a world of two, and
    the factored exchange must have reproduced the plain all-reduce (exchange_selfcheck) -- BASELINE configs[3], SURVEY 8(e)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_bare_env(), SURFEL_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "200000",
                        "--no-cpu-baseline"] + extra, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "without a torchrun environment: launching" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen_by_each_rank"] == [2, 2] and d["config"]["backend"] == "gloo"
    assert d["config"]["frames_per_step"] == 2 * (2 if "--frames-per-rank" in extra else 1)
    if "--exchange" in extra:
        assert d["config"]["gradient_exchange"].startswith("all-reduce of 232")
    else:
        assert d["config"]["exchange_selfcheck"]["ok"] and d["config"]["exchange_selfcheck"]["factored_vs_allreduce_max_rel_err"] < 1e-4
        assert d["config"]["gradient_exchange"].startswith("all-gather")
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_bench_one_rank_rccl_reports_the_exchange():
    """The N > 1 branches of bench.py over the REAL backend (`nccl` = RCCL) on this one GPU (SURFEL_EXCHANGE_SINGLE_RANK=1: a one-rank group
    takes the same calls as an N-rank one), and the fields the first real multi-GPU run has to explain itself with: per-rank step time,
    the EXPOSED exchange time (compute-stream stalls at the wait points), the collectives timed alone with their bandwidths, the predicted
    xGMI wire time, and what RCCL's own log says the communicator is made of."""
    import json, socket, subprocess, sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_bare_env(), SURFEL_EXCHANGE_SINGLE_RANK="1", SURFEL_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("NCCL_DEBUG", None); env.pop("NCCL_DEBUG_FILE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--gaussians", "300000",
                        "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["backend"] == "nccl" and c["world_size_seen_by_each_rank"] == [1] and c["exchange_selfcheck"]["ok"]
    ex = c["exchange"]
    assert len(ex["per_rank_ms_per_step"]) == 1 and ex["per_rank_ms_per_step"][0] > 0
    assert len(ex["exposed_ms_per_step_per_rank"]) == 1 and 0 <= ex["exposed_ms_per_step_max_over_ranks"] < ex["per_rank_ms_per_step"][0]
    iso = ex["isolated_collectives"]
    assert iso["all_gather_colour_gradients"]["bytes_per_rank"] == 300000 * 12 and iso["all_gather_colour_gradients"]["ms"] > 0
    assert iso["all_reduce_rest"]["bytes"] == 300000 * 40 and iso["all_reduce_rest"]["ms"] > 0 and iso["all_reduce_rest"]["algbw_GBs"] > 0
    assert ex["collectives_alone_ms_per_step"] > 0 and ex["hidden_fraction_of_the_collectives"] is not None
    assert "predicted_xgmi" in ex and c["rccl"]["backend"] == "nccl"
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump({"rccl": c["rccl"], "exchange": ex}, open(os.path.join(root, "gpurun_out", "one_rank_rccl_exchange.json"), "w"), indent=1)
    if c["rccl"].get("log") and os.path.exists(c["rccl"]["log"]):   # keep RCCL's own account of the communicator next to it
        open(os.path.join(root, "gpurun_out", "one_rank_rccl.log"), "w").write(open(c["rccl"]["log"], errors="replace").read()[-20000:])


def test_bench_gpus_8_ranks_on_one_gpu():
    """BASELINE configs[3] / [4] shard eight cameras over eight GPUs; no 8-GPU node is available to the tests, but the 8-RANK code path is:
    `python bench.py --gpus 8` (bare: it launches itself) with eight ranks sharing this box's one GPU over gloo and a small scene -- the
    camera batch yawed (k - 3.5) x 5 degrees, the all-gather of eight colour gradients paired with eight cameras, the flat all-reduce, and
    the factored exchange checked against the plain all-reduce on every rank."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_bare_env(), SURFEL_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--gaussians", "60000",
                        "--width", "640", "--height", "360", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["world_size_seen_by_each_rank"] == [8] * 8 and d["config"]["frames_per_step"] == 8
    assert d["config"]["exchange_selfcheck"]["ok"] and d["config"]["parallelism"].startswith("frame-sharded dp8")
    assert d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.parametrize("tile", [(8, 8), (16, 8), (32, 8), (32, 16)])
def test_tile_shape_sweep(tile):
    """BASELINE config 5's tile-size sweep: every shape bins bit-exactly like the oracle run with the same BLOCK_X x BLOCK_Y,
    renders/differentiates within the usual bars.  (Images are NOT identical across shapes, here or in the reference: alpha
    >= 1/255 reaches out to 3.33 sigma for opaque splats while the tile rectangle is cut at 3 sigma, so the fringe is clipped
    at tile granularity.)"""
    from tests.gpu_util import run_hip, run_hip_raw, run_oracle
    P, W, H = 4000, 203, 117          # not a multiple of any tile size
    cam, g = _scene(P, W, H, 17, 4e-3, 8e-2, 2)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=8)
    fwd, bwd = run_oracle(g, cam, bg, 3, dc, da, tile=tile)
    gx, gy = (W + tile[0] - 1) // tile[0], (H + tile[1] - 1) // tile[1]
    assert fwd["ranges"].shape[0] == gx * gy
    raw = run_hip_raw(g, cam, bg, 3, tile=tile)
    _check_binning(raw, fwd)
    out = run_hip(g, cam, bg, 3, dc, da, tile=tile)
    _check_images(out, fwd, f"tile {tile}")
    _check_grads(out, bwd, ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"], f"tile {tile}")
    ref = run_hip(g, cam, bg, 3, dc, da)
    both = (out["radii"] > 0) & (ref["radii"] > 0)   # (an off-screen splat's clamped tile rectangle can be empty for one shape only)
    np.testing.assert_array_equal(out["radii"][both], ref["radii"][both])
    assert (out["radii"] > 0).sum() >= 0.99 * (ref["radii"] > 0).sum()
    assert np.abs(out["color"] - ref["color"]).mean() < 1e-3   # same picture up to the clipped fringe
    # unsupported shapes are refused, not silently replaced
    from streetunveiler_amd._lib import SurfelRasterError
    with pytest.raises(SurfelRasterError):
        run_hip(g, cam, bg, 3, tile=(24, 16))


def test_very_long_tile_lists_and_tiny_images():
    """One tile with a list far longer than a staging round can see (40 k translucent splats over a 24x20 image: > 600 rounds of
    64, 16-bit contributor counts exceeded), and images smaller than a tile / a single pixel."""
    from tests.gpu_util import run_hip, run_hip_raw, run_oracle
    P, W, H = 40000, 24, 20
    cam, g = _scene(P, W, H, 41, 2e-2, 2e-1, 0)
    g["opacities"] = g["opacities"] * 0.02          # nothing saturates: every pixel walks (almost) the whole list
    bg = np.array([0.3, 0.6, 0.9], np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=5)
    fwd, bwd = run_oracle(g, cam, bg, 2, dc, da)
    assert (fwd["ranges"][:, 1] - fwd["ranges"][:, 0]).max() > 20000
    assert fwd["n_contrib"][0].max() > 5000
    _check_binning(run_hip_raw(g, cam, bg, 2), fwd)
    out = run_hip(g, cam, bg, 2, dc, da)
    _check_images(out, fwd, "long lists")
    _check_grads(out, bwd, ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"], "long lists")
    for (w, h) in ((5, 3), (1, 1), (17, 1)):
        cam2, g2 = _scene(300, w, h, 42, 5e-2, 5e-1, 1)
        dc2, da2 = synthetic_upstream_grads(w, h, seed=6)
        f2, b2 = run_oracle(g2, cam2, bg, 1, dc2, da2)
        o2 = run_hip(g2, cam2, bg, 1, dc2, da2)
        np.testing.assert_array_equal(o2["radii"], f2["radii"])
        np.testing.assert_allclose(o2["color"], f2["color"], atol=2e-4)
        np.testing.assert_allclose(o2["allmap"][[0, 1, 2, 3, 4, 6]], f2["allmap"][[0, 1, 2, 3, 4, 6]], atol=2e-3, rtol=2e-3)
        for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"):
            sc = np.abs(b2[k]).max() + 1e-20
            assert np.abs(o2[k].reshape(b2[k].shape) - b2[k]).max() <= 2e-2 * sc, (w, h, k)


def test_randomised_scenes_short_sweep():
    """tools/fuzz_parity.py on a dozen random scenes (sizes, SH degree, tile shape, opacity / scale regimes, precomputed
    colours): bit-exact binning, images within one flipped contributor, at most max(3, 0.1 %) Gaussians with an
    out-of-tolerance gradient."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "12", "2000"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "12/12 scenes within the parity bars" in r.stdout
    # scenes a 250-scene sweep found in round 3: splats hundreds of pixels wide whose centre projects far off-screen -- the moments of dL/dp
    # were taken about that centre and cancelled (dL_dscales rows 2.6e-2 off with identical decisions, the float32 oracle 7e-6); the
    # reference point is now the centre clamped into the image, and the kernels are the more accurate of the two there
    env = dict(os.environ, FUZZ_SEEDS="7021,7063,7082,9057,9061")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py")], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "5/5 scenes within the parity bars" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # cameras in general position (seeds >= 100000: any rotation, the centre up to 200 units out, FoVx unrelated to FoVy) + the two scenes of a
    # 600-scene sweep of them that needed the per-row / non-robust-count readings of the bars (tests/gpu_util.py rows_within, differing pixels)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "10", "100000"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "10/10 scenes within the parity bars" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py")], cwd=root, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FUZZ_SEEDS="100525,100551"))
    assert r.returncode == 0 and "2/2 scenes within the parity bars" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # needle-shaped surfels (seeds >= 400000: half of the Gaussians with an axis ratio of 10 .. 316; the kernels' staged cross products hold
    # the bars up to there -- 250 / 250 -- and start to lose them around 1000 : 1, tools/notes_round5_measured.md)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "10", "400000"], cwd=root, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FUZZ_NEEDLE_MAX_LOG10="2.5"))
    assert r.returncode == 0 and "10/10 scenes within the parity bars" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_extensions_short_sweep():
    """tools/fuzz_extensions.py on a dozen random scenes + the seeds a 1 000-scene sweep needed its float64 fallback for: the 9-channel
    pass, the per-class pass, mask= and the fused activations against the plain operator calls of this build they replace (forward bit
    for bit where the arithmetic is the same, gradients as the sum of the replaced calls' gradients), over random sizes, tile shapes and
    regimes."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_extensions.py"), "12", "5000"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "12/12 scenes consistent" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    for seed in ("3427", "3574", "30703"):   # distortion-only gradients that are float32 cancellation noise: held against the float64 backward instead
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_extensions.py"), "1", seed], cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "1/1 scenes consistent" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_degenerate_parameters():
    """Zero / sub-denormal / gigantic scales, zero quaternions, opacity exactly 0 and 1: same radii, no NaN or Inf on either
    side, images and gradients within the usual bars (scales underflow in training; nothing here may poison a frame)."""
    from tests.gpu_util import run_hip, run_oracle
    P, W, H = 6000, 240, 136
    cam, g = _scene(P, W, H, 77, 3e-3, 5e-2, 2)
    idx = np.random.default_rng(0).permutation(P)
    g["scales"][idx[:60], 0] = 0.0; g["scales"][idx[60:120]] = 0.0
    g["opacities"][idx[120:180]] = 0.0; g["opacities"][idx[180:240]] = 1.0
    g["rotations"][idx[240:300]] = 0.0
    g["scales"][idx[300:360]] = 1e-12; g["scales"][idx[360:420]] = 50.0
    bg = np.zeros(3, np.float32)
    dc, da = synthetic_upstream_grads(W, H, seed=3)
    fwd, bwd = run_oracle(g, cam, bg, 3, dc, da)
    out = run_hip(g, cam, bg, 3, dc, da)
    np.testing.assert_array_equal(out["radii"], fwd["radii"])
    names = ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"]
    for a in [out["color"], out["allmap"], fwd["color"], fwd["allmap"]] + [out[n] for n in names] + [bwd[n] for n in names]:
        assert np.isfinite(np.asarray(a)).all()
    _check_images(out, fwd, "degenerate")
    _check_grads(out, bwd, names, "degenerate")


def test_non_finite_parameters_do_not_spread():
    """A diverging training run hands the operator NaN / Inf parameters.  A Gaussian whose centre, scales or rotation are not finite is culled
    (radius 0, no gradient); one whose opacity or SH coefficients are not finite still renders (`min(0.99, NaN)` is 0.99 here as in the
    reference's CUDA) and may carry non-finite gradients in ITS OWN rows -- in every case the images stay finite and no other Gaussian's
    gradient row is touched."""
    from tests.gpu_util import run_hip
    P, W, H = 6000, 240, 136
    cam, g = _scene(P, W, H, 3, 3e-3, 5e-2, 2)
    dc, da = synthetic_upstream_grads(W, H, seed=3)
    idx = torch.arange(0, P, 50)
    healthy = np.ones(P, bool); healthy[idx.numpy()] = False
    names = ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D")
    for field, val in [("means3D", float("nan")), ("means3D", float("inf")), ("scales", float("nan")), ("scales", float("inf")), ("rotations", float("nan")),
                       ("opacities", float("nan")), ("opacities", float("inf")), ("shs", float("nan"))]:
        gg = {k: v.clone() for k, v in g.items()}
        if field == "means3D": gg[field][idx, 2] = val
        elif field == "shs": gg[field][idx, 0, 0] = val
        else: gg[field][idx, 0] = val
        out = run_hip(gg, cam, [0.1, 0.2, 0.3], 3, dc, da)
        tag = f"{field} = {val}"
        assert np.isfinite(out["color"]).all() and np.isfinite(out["allmap"]).all(), tag
        bad = {k: ~np.isfinite(np.asarray(out[k]).reshape(P, -1)).all(1) for k in names}
        assert not any(b[healthy].any() for b in bad.values()), f"{tag}: a healthy Gaussian has a non-finite gradient row"
        if field in ("means3D", "scales", "rotations"):
            assert not out["radii"][idx.numpy()].any() and not any(b.any() for b in bad.values()), f"{tag}: the poisoned Gaussians must be culled"
            for k in names:
                assert not np.asarray(out[k]).reshape(P, -1)[idx.numpy()].any(), f"{tag}: {k} of a culled Gaussian is not zero"


def test_wide_frame_with_few_gaussians_and_counter_variant_errors():
    """(a) One Gaussian in a frame 750 tile columns wide: pass X's [columns][blocks] histogram lives in the geometry scratch, which is
    sized for the widest frame the binning accepts (it used to be sized by P alone: BUFFER_TOO_SMALL for P = 1 beyond 640 columns).
    (b) The counting variant of the forward blend exists for the 16x16 tile with 3 or 6 channels: any other request is refused with
    SR_ERR_UNSUPPORTED instead of returning zeros that look like measurements."""
    from streetunveiler_amd import _lib
    from tests.gpu_util import DEV, run_hip_raw, run_oracle
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import settings_for
    W, H = 12000, 32
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(3, W, H, seed=2, scale_lo=1e-3, scale_hi=2e-3)
    g = {k: v[:1].contiguous() for k, v in g.items()}
    g["means3D"][0] = torch.tensor([0.0, 0.0, 5.0])
    bg = np.zeros(3, np.float32)
    fwd, _ = run_oracle(g, cam, bg, 3)
    raw = run_hip_raw(g, cam, bg, 3)
    assert raw["D"] == fwd["num_rendered"] and raw["D"] >= 1
    np.testing.assert_array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    np.testing.assert_allclose(raw["color"], fwd["color"], atol=1e-4)
    # (b)
    cam2, g2 = _scene(500, 96, 64, 3, 5e-3, 5e-2)
    s = settings_for(cam2, [0, 0, 0], 1)
    d = {k: v.to(DEV) for k, v in g2.items()}
    counters = torch.zeros(16, dtype=torch.int64, device=DEV)
    kw = dict(means3D=d["means3D"], means2D=torch.zeros(500, 3, device=DEV), shs=d["shs"], opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"])
    GaussianRasterizer(s, blend_counters=counters)(**kw)
    torch.cuda.synchronize()
    assert counters[0] > 0 and counters[7] <= counters[1] and counters[9] <= counters[8] <= counters[10]
    with pytest.raises(_lib.SurfelRasterError, match="blend_counters"):
        GaussianRasterizer(s, tile=(32, 16), blend_counters=counters)(**kw)


def test_forward_only_mode_is_bit_identical_and_writes_no_backward_state():
    """SR_FLAG_FORWARD_ONLY: the reference's inference callers run the operator under torch.no_grad() [REF /root/reference/render.py:68;
    /root/reference/utils/mesh_utils.py:82-100] -- the shim then tells the library that no backward follows.  color / allmap / radii are
    bit-identical to the training forward (3 channels with SHs and with precomputed colours, 6 / 9 channels, both forward mappings, a second
    tile shape); the image state comes back empty; a call with grad enabled is NOT forward-only and still backpropagates."""
    from diff_surfel_rasterization import GaussianRasterizer, _C
    from tests.gpu_util import DEV, settings_for
    W, H, P = 330, 200, 30_000
    cam, g = _scene(P, W, H, 21, 1e-3, 2e-2, cam_index=2)
    d = {k: v.to(DEV) for k, v in g.items()}
    e = torch.empty(0, device=DEV)
    s = settings_for(cam, [0.1, 0.3, 0.2], 3)
    cols6 = torch.rand(P, 6, generator=torch.Generator().manual_seed(3)).to(DEV)
    bg6 = torch.rand(6, generator=torch.Generator().manual_seed(4)).to(DEV)
    cases = [dict(sh=d["shs"], col=e, bg=s.bg, kw={}), dict(sh=e, col=cols6[:, :3].contiguous(), bg=s.bg, kw={}),
             dict(sh=e, col=cols6, bg=bg6, kw={}), dict(sh=d["shs"], col=cols6, bg=torch.cat([s.bg, bg6]), kw={}),
             dict(sh=d["shs"], col=e, bg=s.bg, kw=dict(row_mapped=True)), dict(sh=d["shs"], col=e, bg=s.bg, kw=dict(row_mapped=False)),
             dict(sh=d["shs"], col=e, bg=s.bg, kw=dict(tile=(32, 16))), dict(sh=d["shs"], col=e, bg=s.bg, kw=dict(tile=(8, 8)))]
    for c in cases:
        run = lambda fo: _C.rasterize_gaussians(c["bg"], d["means3D"], c["col"], d["opacities"], d["scales"], d["rotations"], 1.0, e, s.viewmatrix, s.projmatrix,
                                                s.tanfovx, s.tanfovy, H, W, c["sh"], 3, s.campos, False, False, forward_only=fo, **c["kw"])
        D0, color0, allmap0, radii0, _, _, img0 = run(False)
        D1, color1, allmap1, radii1, _, _, img1 = run(True)
        assert D0 == D1 and D0 > 50_000
        assert torch.equal(color0, color1) and torch.equal(allmap0, allmap1) and torch.equal(radii0, radii1), c["kw"]
        assert img1.numel() == 0 and img0.numel() > 0
    # through the operator: no_grad -> forward-only; grad mode with a leaf that requires grad -> the training forward
    seen = []
    real = _C.rasterize_gaussians
    spy = lambda *a, **k: (seen.append(bool(k.get("forward_only"))), real(*a, **k))[1]
    _C.rasterize_gaussians = spy
    try:
        t = {k: v.clone().requires_grad_() for k, v in d.items()}
        m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
        call = lambda: GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        with torch.no_grad():
            c_ng, _, a_ng = call()
        c_g, _, a_g = call()
        c_det, _, _ = GaussianRasterizer(s)(means3D=d["means3D"], means2D=torch.zeros(P, 3, device=DEV), shs=d["shs"], opacities=d["opacities"],
                                            scales=d["scales"], rotations=d["rotations"])   # grad mode on, but nothing requires grad
    finally:
        _C.rasterize_gaussians = real
    assert seen == [True, False, True]
    assert torch.equal(c_ng, c_g.detach()) and torch.equal(a_ng, a_g.detach()) and torch.equal(c_det, c_ng)
    (c_g.sum() + a_g.sum()).backward()
    assert t["means3D"].grad is not None and torch.isfinite(t["means3D"].grad).all() and t["shs"].grad.abs().sum() > 0


@pytest.mark.parametrize("tile", [None, (8, 8)])
def test_constant_extra_colours_skip_their_gradient_sums(tile):
    """SR_FLAG_NO_PRECOMP_COLOR_GRAD: when colors_precomp[P,6] / extra_colors does not require grad (render_semantic's one-hot class channels
    are constants) the shim tells K7 not to form dL/dcolors_precomp.  Every other gradient is bit-identical to the run that does form it
    (the sums of the other slots go through the same reduction tree), for the 6- and the 9-channel pass; on the reference tile that is the
    kXG = false instantiation, on another shape the flag changes nothing."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import DEV, settings_for
    P, W, H = 6000, 240, 136
    cam, g = _scene(P, W, H, 61, 3e-3, 5e-2, 3)
    six = torch.rand(P, 6, generator=torch.Generator().manual_seed(1)).to(DEV)
    gen = torch.Generator().manual_seed(2)
    for nc in (6, 9):
        bg = np.linspace(0.1, 0.7, nc).astype(np.float32)
        s = settings_for(cam, bg, 3 if nc == 9 else 0)
        w_c = torch.randn(nc, H, W, generator=gen).to(DEV); w_a = torch.randn(7, H, W, generator=gen).to(DEV)

        def run(extra_needs_grad):
            t = {k: v.to(DEV).clone().requires_grad_() for k, v in g.items()}
            m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
            ex = six.clone().requires_grad_(extra_needs_grad)
            kw = dict(shs=t["shs"], extra_colors=ex) if nc == 9 else dict(colors_precomp=ex)
            c, _, a = GaussianRasterizer(s, tile=tile)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], **kw)
            ((c * w_c).sum() + (a * w_a).sum()).backward()
            grads = {k: t[k].grad for k in ("means3D", "opacities", "scales", "rotations")}
            grads["m2d"] = m2d.grad
            if nc == 9:
                grads["shs"] = t["shs"].grad
            return grads, ex.grad

        with_g, gx = run(True)
        without, none = run(False)
        assert gx is not None and float(gx.abs().max()) > 0 and none is None
        for k in with_g:
            assert torch.equal(with_g[k], without[k]), (nc, tile, k)


def test_tile_auto_on_the_references_own_frames_against_the_oracle():
    """`GaussianRasterizer(tile="auto")` on a frame of the reference's documented runs (`-r 4`: 480x320 [REF README.md:195-207]) picks the 8x8
    tile -- 600 tiles of 16x16 cannot fill the GPU -- and what it renders is held to the oracle run WITH THAT TILE: duplicate count, sorted
    list and ranges bit-exact, images and gradients by the usual float32-oracle bars; and to the 16x16 render of the same frame (the tile
    shape is invisible in the results beyond float summation order)."""
    from diff_surfel_rasterization import GaussianRasterizer, resolve_tile
    from tests.gpu_util import DEV, assert_close_frac, assert_grads_close, check_allmap, run_hip, run_hip_raw, run_oracle, settings_for
    from tests.bars import bar
    W, H, P = 480, 320, 150_000
    assert resolve_tile("auto", W, H) == (8, 8)
    cam = synthetic_camera(W, H, index=5)
    g = synthetic_gaussians(P, W, H, seed=21, scale_lo=1e-3, scale_hi=8e-3)
    dc, da = synthetic_upstream_grads(W, H, seed=3)
    bg = np.array([0.2, 0.1, 0.0], np.float32)
    fwd, bwd = run_oracle(g, cam, bg, 3, dc, da, tile=(8, 8))
    raw = run_hip_raw(g, cam, bg, 3, tile=(8, 8))
    assert raw["D"] == fwd["num_rendered"] and raw["D"] > 300_000
    np.testing.assert_array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
    np.testing.assert_array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])
    # through the rasterizer with tile="auto"
    t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
    m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    r = GaussianRasterizer(settings_for(cam, bg, 3), tile="auto")
    assert r.tile == (8, 8)
    color, radii, allmap = r(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    ((color * dc.to(DEV)).sum() + (allmap * da.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(radii.cpu().numpy(), fwd["radii"])
    assert_close_frac(color.detach().cpu().numpy(), fwd["color"], bar("oracle32_image_atol"), bar("oracle32_image_atol"), bar("oracle32_image_bad_frac_small"),
                      bar("oracle32_image_hard"), "tile auto color")
    check_allmap(allmap.detach().cpu().numpy(), fwd["allmap"], "tile auto")
    for name, leaf in [("dL_dmeans3D", t["means3D"]), ("dL_dopacity", t["opacities"]), ("dL_dscales", t["scales"]), ("dL_drotations", t["rotations"]),
                       ("dL_dsh", t["shs"]), ("dL_dmeans2D", m2)]:
        assert_grads_close(leaf.grad.cpu().numpy(), bwd[name], bar("oracle32_grad_rel"), "tile auto " + name)
    # NOT compared with the 16x16 render: the reference's algorithm truncates a splat at the TILES its 3-sigma bounding box touches, and an
    # opaque splat's alpha >= 1/255 footprint reaches up to 3.3 sigma -- with smaller tiles fewer of those fringe pixels lie in a listed
    # tile.  On this scene 2.7 % of the pixels differ by more than 1e-4 between the two tile shapes (measured; up to 4e-2).  `tile="auto"`
    # is therefore an OPTION that changes the rendering at the reference's own truncation fringe; the default path for small frames is the
    # cooperative backward on the reference's 16x16 lists (test_cooperative_backward_equals_the_one_wave_backward).
    ref16 = run_hip(g, cam, bg, 3, dc, da)
    assert np.array_equal(ref16["radii"], radii.cpu().numpy())
    differing = float((np.abs(color.detach().cpu().numpy() - ref16["color"]) > 1e-4).mean())
    assert 0.0 < differing < 0.1, f"{differing:.3f} of the colour elements differ between 8x8 and 16x16 tiles"


@pytest.mark.parametrize("size", [(480, 320, 150_000), (1280, 720, 300_000)])
def test_cooperative_backward_equals_the_one_wave_backward(size):
    """The blend backward as four quadrant waves per 16x16 tile (picked by itself below 2 600 tiles: the reference's `-r 4` frames) against the
    one-wave-per-tile kernel on the same tile lists: forward bit-identical (it is the same forward), every gradient equal up to the order of a
    four-term float sum -- and held to the oracle by the same bars as the default path; bit-identical reruns."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import DEV, assert_grads_close, run_oracle, settings_for
    from tests.bars import bar
    W, H, P = size
    cam = synthetic_camera(W, H, index=2)
    g = synthetic_gaussians(P, W, H, seed=33, scale_lo=1e-3, scale_hi=8e-3)
    dc, da = synthetic_upstream_grads(W, H, seed=4)
    bg = np.array([0.1, 0.0, 0.2], np.float32)

    def step(kernel):
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
        c, r, a = GaussianRasterizer(settings_for(cam, bg, 3), backward_kernel=kernel)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"],
                                                                                      scales=t["scales"], rotations=t["rotations"])
        ((c * dc.to(DEV)).sum() + (a * da.to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        out = dict(dL_dmeans3D=t["means3D"].grad, dL_dopacity=t["opacities"].grad, dL_dscales=t["scales"].grad, dL_drotations=t["rotations"].grad, dL_dsh=t["shs"].grad, dL_dmeans2D=m2.grad)
        return {k: v.cpu().numpy() for k, v in out.items()}
    # the forward of the same switch: four quadrant waves sharing one staging vs two band waves -- images and state bit for bit
    from tests.gpu_util import run_hip_raw
    import diff_surfel_rasterization._C as _Cmod
    s_ = settings_for(cam, bg, 3); e_ = torch.empty(0, device=DEV); d_ = lambda k: g[k].to(DEV)
    fw = {}
    for kernel in ("one_wave", "coop"):
        D_, col_, am_, rad_, geom_, bin_, img_ = _Cmod.rasterize_gaussians(s_.bg, d_("means3D"), e_, d_("opacities"), d_("scales"), d_("rotations"), 1.0, e_, s_.viewmatrix, s_.projmatrix,
                                                                            s_.tanfovx, s_.tanfovy, H, W, d_("shs"), 3, s_.campos, False, False, backward_kernel=kernel)
        iv = _Cmod.image_view(img_, W, H)
        fw[kernel] = (col_.cpu(), am_.cpu(), rad_.cpu(), iv["final_T"].cpu(), iv["n_contrib"].cpu())
    for a_, b_, what in zip(fw["one_wave"], fw["coop"], ("color", "allmap", "radii", "final_T", "n_contrib")):
        assert torch.equal(a_, b_), f"cooperative forward: {what} differs from the band kernel's"
    one, coop, coop2, auto = step("one_wave"), step("coop"), step("coop"), step(None)
    few = ((W + 15) // 16) * ((H + 15) // 16) < 2600
    _, bwd = run_oracle(g, cam, bg, 3, dc, da)
    for k in one:
        assert np.array_equal(coop[k], coop2[k]), f"{k}: the cooperative backward is not deterministic"
        assert np.array_equal(auto[k], coop[k] if few else one[k]), f"{k}: the default picks the {'cooperative' if few else 'one-wave'} kernel at {W}x{H}"
        scale = np.abs(one[k]).max() + 1e-30
        # (HIP vs HIP: summation order only)
        assert np.abs(coop[k] - one[k]).max() <= bar("class_grads_vs_operator") * scale, f"{k}: cooperative vs one-wave differ by {np.abs(coop[k] - one[k]).max() / scale:.2e} of the tensor scale"
        assert_grads_close(coop[k], bwd[k], bar("oracle32_grad_rel"), "coop " + k)


@pytest.mark.parametrize("size", [(480, 320, 150_000), (1280, 720, 300_000)])
def test_row_mapped_backward_equals_the_one_wave_backward(size, kernel="rows"):
    """The row-mapped blend pair (SR_FLAG_ROW_BACKWARD: the forward keeps its hit masks per (entry, 4x4 cell), the backward's four 16-lane rows walk
    their own cell's list and add their per-step sums to the entry's record row in LDS) against the default pair on the same tile lists: forward
    bit-identical, every gradient equal up to the order of the additions, held to the oracle by the same bar, bit-identical reruns."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import DEV, assert_grads_close, run_oracle, settings_for
    from tests.bars import bar
    W, H, P = size
    cam = synthetic_camera(W, H, index=2)
    g = synthetic_gaussians(P, W, H, seed=33, scale_lo=1e-3, scale_hi=8e-3)
    dc, da = synthetic_upstream_grads(W, H, seed=4)
    bg = np.array([0.1, 0.0, 0.2], np.float32)

    def step(k_):
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
        c, r, a = GaussianRasterizer(settings_for(cam, bg, 3), backward_kernel=k_)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"],
                                                                                  scales=t["scales"], rotations=t["rotations"])
        ((c * dc.to(DEV)).sum() + (a * da.to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        out = dict(color=c.detach(), allmap=a.detach(), radii=r, dL_dmeans3D=t["means3D"].grad, dL_dopacity=t["opacities"].grad, dL_dscales=t["scales"].grad,
                   dL_drotations=t["rotations"].grad, dL_dsh=t["shs"].grad, dL_dmeans2D=m2.grad)
        return {k: v.cpu().numpy() for k, v in out.items()}
    one, rows, rows2 = step("one_wave"), step(kernel), step(kernel)
    _, bwd = run_oracle(g, cam, bg, 3, dc, da)
    for k in ("color", "allmap", "radii"):
        assert np.array_equal(one[k], rows[k]), f"row-mapped pair: {k} differs from the default forward's"
    for k in [k_ for k_ in one if k_.startswith("dL_")]:
        assert np.array_equal(rows[k], rows2[k]), f"{k}: the row-mapped backward is not deterministic"
        scale = np.abs(one[k]).max() + 1e-30
        assert np.abs(rows[k] - one[k]).max() <= bar("class_grads_vs_operator") * scale, f"{k}: row-mapped vs one-wave differ by {np.abs(rows[k] - one[k]).max() / scale:.2e} of the tensor scale"
        assert_grads_close(rows[k], bwd[k], bar("oracle32_grad_rel"), kernel + " " + k)
