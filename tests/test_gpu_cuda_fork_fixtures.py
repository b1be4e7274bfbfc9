"""HIP kernels AND the CPU oracle against vectors rendered by the reference's real CUDA rasterizer.

The reference's native module `diff_surfel_rasterization._C` is an empty submodule in /root/reference (.gitmodules:9-12; importer
gaussian_renderer/__init__.py:11, call :129-138), so nothing in this repository can produce such a vector: K1's geometry, K6, K7 and K8 are
pinned by the builder's own oracle only ("parity unpinned").  `tools/make_cuda_fixtures.py` is the stand-alone kit for
whoever has the fork: it writes tests/golden/cuda_fork_<scene>.npz (inputs, settings, upstream gradients, color / radii / allmap, every
input gradient).  This file consumes them:

  * present  -> kernels vs fixture and oracle vs fixture: radii bit-exact; colour + the seven aux maps within 1e-4 (1 + |v|) at every
               ROBUST pixel (no decision of the free-running float64 checker within float32 noise of its threshold), the loose 2e-2 cap
               elsewhere; gradient rows by the bars of tests/gpu_util.py (STRICT_ROW_BARS on robust Gaussians, 5e-2 of the tensor scale
               on the rest).  On a mismatch the message carries the named-switch matrix: which build of include/surfel_switches.h (oracle
               side, CPU) agrees with the fixture -- the switch to flip in the shipped build.
  * absent   -> SKIPPED, loudly: parity stays "partial".
  * always (on a GPU) -> the kit itself is exercised: the script renders its scenes through THIS repository's drop-in package
               (`--self-test-with-drop-in`), and the comparison code runs on those files (HIP vs HIP is trivially equal; the oracle leg is real).
"""
import glob
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import gpu_util as gu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
KIT = os.path.join(ROOT, "tools", "make_cuda_fixtures.py")
GRADS = [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
         ("shs", "dL_dsh"), ("colors_precomp", "dL_dcolors"), ("means2D", "dL_dmeans2D")]


def load(path):
    d = dict(np.load(path))
    fx = dict(name=os.path.basename(path)[:-4], P=d["means3D"].shape[0], W=int(d["image_width"]), H=int(d["image_height"]),
              deg=int(d["sh_degree"]), tanfovx=float(d["tanfovx"]), tanfovy=float(d["tanfovy"]), scale_modifier=float(d["scale_modifier"]))
    fx["in"] = {k: d[k] for k in ("means3D", "scales", "rotations", "opacities", "shs", "colors_precomp") if k in d}
    fx["cam"] = {k: d[k] for k in ("viewmatrix", "projmatrix", "campos", "bg")}
    fx["up"] = (d["dL_dcolor"], d["dL_dallmap"])
    fx["out"] = dict(color=d["out_color"], radii=d["out_radii"].astype(np.int32), allmap=d["out_allmap"])
    for src, name in GRADS:
        if "grad_" + src in d:
            fx["out"][name] = d["grad_" + src]
    return fx


def oracle_kwargs(fx, **extra):
    return dict(viewmatrix=fx["cam"]["viewmatrix"], projmatrix=fx["cam"]["projmatrix"], campos=fx["cam"]["campos"], bg=fx["cam"]["bg"],
                image_width=fx["W"], image_height=fx["H"], sh_degree=fx["deg"], scale_modifier=fx["scale_modifier"],
                tanfovx=fx["tanfovx"], tanfovy=fx["tanfovy"], **extra)


def run_oracle(fx, so=None, **extra):
    so = so or gu.so
    i = fx["in"]
    fwd = so.rasterize_forward(i["means3D"], i["opacities"], i["scales"], i["rotations"], shs=i.get("shs"), colors_precomp=i.get("colors_precomp"),
                               **oracle_kwargs(fx, **extra))
    return fwd, so.rasterize_backward(fwd, *fx["up"])


def run_kernels(fx):
    """The fixture's inputs through the drop-in operator, called as the reference calls it."""
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = gu.DEV
    t = {k: torch.tensor(v, device=dev).requires_grad_() for k, v in fx["in"].items()}
    means2D = torch.zeros(fx["P"], 3, device=dev, requires_grad=True)
    c = fx["cam"]
    s = GaussianRasterizationSettings(image_height=fx["H"], image_width=fx["W"], tanfovx=fx["tanfovx"], tanfovy=fx["tanfovy"],
                                      bg=torch.tensor(c["bg"], device=dev), scale_modifier=fx["scale_modifier"],
                                      viewmatrix=torch.tensor(c["viewmatrix"], device=dev), projmatrix=torch.tensor(c["projmatrix"], device=dev),
                                      sh_degree=fx["deg"], campos=torch.tensor(c["campos"], device=dev), prefiltered=False, debug=False)
    color, radii, allmap = GaussianRasterizer(raster_settings=s)(means3D=t["means3D"], means2D=means2D, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
                                                                 opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    ((color * torch.tensor(fx["up"][0], device=dev)).sum() + (allmap * torch.tensor(fx["up"][1], device=dev)).sum()).backward()
    torch.cuda.synchronize()
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy().astype(np.int32), allmap=allmap.detach().cpu().numpy())
    for src, name in GRADS:
        v = means2D if src == "means2D" else t.get(src)
        if v is not None:
            out[name] = (torch.zeros_like(v) if v.grad is None else v.grad).cpu().numpy()
    return out


def compare(fx, got, who, fwd64, bwd64, margins):
    """`got` (kernels' or oracle's outputs) against the fixture, split by the float64 checker's robust classification.  -> list of failures."""
    ref, bad = fx["out"], []
    if not np.array_equal(got["radii"], ref["radii"]):
        bad.append(f"{who}: radii differ on {int((got['radii'] != ref['radii']).sum())} of {fx['P']} Gaussians")
        return bad   # (another footprint: everything downstream differs)
    rob_px = margins["pixel"] > 1.0
    rob_med = rob_px & (margins["median"] > 1.0)
    for name, a, b, mask in [("color", got["color"], ref["color"], rob_px)] + \
                            [(f"allmap[{c}]", got["allmap"][c], ref["allmap"][c], rob_med if c == 5 else rob_px) for c in range(7)]:
        err = np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b))
        m = np.broadcast_to(mask, err.shape)
        over = err - np.broadcast_to(margins.get("value_noise", 0.0), err.shape)
        if over[m].max(initial=0.0) > gu.BARS["robust_pixel"]:
            bad.append(f"{who}: {name} off by {err[m].max():.3e} of (1 + |v|) at a robust pixel ({int((over[m] > gu.BARS['robust_pixel']).sum())} pixels over 1e-4)")
        if name != "allmap[5]" and err[~m].max(initial=0.0) > gu.BARS["nonrobust_pixel_cap"]:
            bad.append(f"{who}: {name} off by {err[~m].max():.3e} at a non-robust pixel")
    vis = ref["radii"] > 0
    rob_g = vis & (margins["gaussian"] > 1.0)
    for key, (p999_bar, max_bar) in gu.STRICT_ROW_BARS.items():
        if key not in got or key not in ref:
            continue
        P = ref[key].shape[0]
        r = np.asarray(ref[key], np.float64).reshape(P, -1); a = np.asarray(got[key], np.float64).reshape(P, -1)
        if np.abs(a[~vis]).any():
            bad.append(f"{who}: {key} non-zero on an invisible Gaussian")
        e = gu.row_errors(a, r, np.ones(P, bool))
        if key in ("dL_dscales", "dL_drotations"):
            p999_bar, max_bar = 2e-3, 6e-2     # plain row metric against a float32 reference (gpu_util.assert_strict_parity, scene=None)
        # (both sides are float32 here -- the fixture carries the CUDA kernels' own rounding, atomics in arbitrary order -- so the p99.9
        # bar is twice the one against the float64 arbiter)
        if not gu.rows_within(e[rob_g], 2.0 * p999_bar, max_bar):
            er = e[rob_g]
            bad.append(f"{who}: {key} robust rows p99.9 {np.quantile(er, 0.999):.2e} (bar {2 * p999_bar:.1e}), max {er.max():.2e} (bar {max_bar:.1e})")
        loose = np.abs(a - r).max(1) / (np.abs(r).max() + 1e-30)
        if loose[vis & ~rob_g].max(initial=0.0) > gu.BARS["nonrobust_row_cap"]:
            bad.append(f"{who}: {key} a non-robust row is off by {loose[vis & ~rob_g].max():.2e} of the tensor scale")
    return bad


def switch_matrix(fx):
    """Which named switch of include/surfel_switches.h (oracle builds, CPU) reproduces the fixture: {name: #failures}."""
    from oracle import surfel_oracle as so_default
    from streetunveiler_amd.build import VARIANTS
    import importlib
    res = {}
    fwd64, bwd64, margins = None, None, None
    for name in ["shipped"] + sorted(VARIANTS):
        try:
            if name == "shipped":
                so = so_default
            else:
                so_default.build_variant(name)
                so = _variant_module(name)
            f64 = so.rasterize_forward(fx["in"]["means3D"], fx["in"]["opacities"], fx["in"]["scales"], fx["in"]["rotations"], shs=fx["in"].get("shs"),
                                       colors_precomp=fx["in"].get("colors_precomp"), **oracle_kwargs(fx, f64=True))
            mg = so.render_margins(f64, f64=True)
            fwd, bwd = run_oracle(fx, so)
            got = dict(color=fwd["color"], radii=fwd["radii"], allmap=fwd["allmap"], **{k: v for k, v in bwd.items() if k in gu.STRICT_ROW_BARS})
            res[name] = len(compare(fx, got, name, f64, None, mg))
        except Exception as e:   # noqa: BLE001 -- the matrix is a diagnostic: a variant that does not build is reported, not raised
            res[name] = f"error: {e}"
    return res


def _variant_module(name):
    """A second instance of oracle.surfel_oracle bound to the variant's shared object."""
    import importlib.util
    from oracle import surfel_oracle as so
    spec = importlib.util.spec_from_file_location(f"surfel_oracle_{name}", so.__file__)
    mod = importlib.util.module_from_spec(spec)
    os.environ["SURFEL_ORACLE_LIB"] = so.variant_path(name)
    try:
        spec.loader.exec_module(mod)
        mod.lib()
    finally:
        os.environ.pop("SURFEL_ORACLE_LIB", None)
    return mod


def check_fixture(path, with_matrix=True):
    fx = load(path)
    i = fx["in"]
    fwd64 = gu.so.rasterize_forward(i["means3D"], i["opacities"], i["scales"], i["rotations"], shs=i.get("shs"), colors_precomp=i.get("colors_precomp"),
                                    **oracle_kwargs(fx, f64=True))
    margins = gu.so.render_margins(fwd64, f64=True)
    fwd, bwd = run_oracle(fx)
    oracle_out = dict(color=fwd["color"], radii=fwd["radii"], allmap=fwd["allmap"], **{k: v for k, v in bwd.items() if k in gu.STRICT_ROW_BARS})
    bad = compare(fx, run_kernels(fx), "kernels", fwd64, None, margins) + compare(fx, oracle_out, "oracle", fwd64, None, margins)
    if bad and with_matrix:
        bad.append("named-switch matrix (failures per oracle build; 0 = that build reproduces the fixture): %r" % switch_matrix(fx))
    assert not bad, fx["name"] + ":\n  " + "\n  ".join(bad)
    return fx


FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "cuda_fork_*.npz")))


@pytest.mark.gpu
def test_cuda_fork_fixtures():
    if not FIXTURES:
        pytest.skip("NO CUDA-FORK FIXTURES (tests/golden/cuda_fork_*.npz): K1 geometry, K6, K7 and K8 are pinned by the repository's own oracle only "
                    "-- parity stays 'partial'.  Someone with the reference's CUDA rasterizer: `python tools/make_cuda_fixtures.py --out tests/golden` "
                    "(stand-alone, numpy + torch + the fork), then re-run this test.")
    for path in FIXTURES:
        check_fixture(path)


@pytest.mark.gpu
def test_the_fixture_kit_end_to_end_through_the_drop_in_package(tmp_path):
    """The kit's script and this file's comparison, exercised on every scene of the kit with the drop-in package in the fork's place: the
    script runs as a maintainer would run it (a subprocess, nothing imported from it), the files load, HIP equals the file (it wrote it),
    and the oracle is held to the file by the same bars a real fixture would apply."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, KIT, "--self-test-with-drop-in", "--out", str(tmp_path)], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(glob.glob(os.path.join(str(tmp_path), "selftest_*.npz")))
    assert len(files) >= 7, r.stderr
    names = set()
    for path in files:
        fx = check_fixture(path, with_matrix=False)
        names.add(fx["name"].replace("selftest_", ""))
        if fx["name"].endswith("culled"):
            assert not (fx["out"]["radii"] > 0).any() and not np.abs(fx["out"]["dL_dmeans3D"]).any()
    assert {"small", "posed", "clones", "precomp", "culled", "nonunit", "modifier"} <= names
