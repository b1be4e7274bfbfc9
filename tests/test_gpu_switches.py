"""The named compile-time switches of include/surfel_switches.h (SURVEY.md Appendix A's (!) items).  The shipped build has every switch
at upstream's value (sr_build_switches() == 0); every non-default value is BUILT on demand (kernels: streetunveiler_amd.build
build_variant -> lib/variants/<name>/; oracle with the same -D: oracle.surfel_oracle.build_variant -> oracle/variants/<name>/) and
checked -- kernels against oracle under that value (tests/switch_worker.py, one process per build: SURFEL_RASTER_LIB /
SURFEL_ORACLE_LIB), and against the shipped build to show that the switch does what its name says."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streetunveiler_amd import build as sb  # noqa: E402

BITS = {"default": 0, "tightbbox": 1, "detach_weight": 2, "no_radius_floor": 4, "median_plain_index": 8, "proxy_view_depth": 16,
        "backward_wh_from_size": 32, "pz_zero_through_filter": 64}


def test_switch_list_is_shared_by_kernels_and_oracle():
    """One header, included by both sides; every variant of the build script flips exactly one macro that header defines."""
    hdr = open(os.path.join(ROOT, "include", "surfel_switches.h")).read()
    assert '#include "../../include/surfel_switches.h"' in open(os.path.join(ROOT, "streetunveiler_amd", "csrc", "common.h")).read()
    assert '#include "../include/surfel_switches.h"' in open(os.path.join(ROOT, "oracle", "surfel_oracle.c")).read()
    assert set(sb.VARIANTS) == set(BITS) - {"default"}
    for name, defines in sb.VARIANTS.items():
        assert len(defines) == 1 and defines[0].startswith("-DSR_")
        macro = defines[0][2:].split("=")[0]
        assert f"#ifndef {macro}\n#define {macro} " in hdr, macro
    from streetunveiler_amd import _lib
    assert set(_lib.SWITCH_BITS) == set(BITS.values()) - {0}


def _run(name, tmp_path):
    env = dict(os.environ)
    env.pop("SURFEL_RASTER_LIB", None); env.pop("SURFEL_ORACLE_LIB", None)
    if name != "default":
        from oracle import surfel_oracle as so
        # both idempotent: nothing is compiled when the variant is newer than every source (pre-built in the container by
        # `python -m streetunveiler_amd.build --variant all`, the .so travels); hipcc and gcc are on the GPU box too
        lib, oracle = sb.build_variant(name), so.build_variant(name)
        env.update(SURFEL_RASTER_LIB=lib, SURFEL_ORACLE_LIB=oracle)
    out = os.path.join(str(tmp_path), name + ".npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), str(BITS[name]), out], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{name}: " + r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    return _run("default", tmp_path_factory.mktemp("switch_default"))


def _differs(a, b, rel=1e-3):
    return np.abs(a - b).max() > rel * (np.abs(b).max() + 1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(sb.VARIANTS))
def test_switch_variant_against_its_oracle_and_against_the_shipped_build(name, shipped, tmp_path):
    v = _run(name, tmp_path)   # (the worker has already required: kernels == oracle of this build, on four scenes)
    d = shipped
    same_forward = lambda: np.array_equal(v["A_radii"], d["A_radii"]) and np.array_equal(v["A_color"], d["A_color"]) and np.array_equal(v["A_allmap"], d["A_allmap"])
    if name == "tightbbox":            # extent follows the opacity: translucent splats shrink, opaque ones (opacity -> 1) keep 3 sigma
        vis = (v["A_radii"] > 0) & (d["A_radii"] > 0)
        assert (v["A_radii"][vis] <= d["A_radii"][vis]).all() and (v["A_radii"][vis] < d["A_radii"][vis]).mean() > 0.2
    elif name == "no_radius_floor":    # ceil(3 * 0.707) = 3 px is the floor: without it sub-pixel splats drop to 1 px
        vis = (v["A_radii"] > 0) & (d["A_radii"] > 0)
        assert (d["A_radii"][vis] >= 3).all() and (v["A_radii"][vis] < 3).any() and (v["A_radii"][vis] <= d["A_radii"][vis]).all()
    elif name == "detach_weight":      # backward only: the distortion no longer pulls on the blend weights
        assert same_forward() and _differs(v["A_dL_dopacity"], d["A_dL_dopacity"], rel=1e-6)   # (a small term next to the colour gradients of this scene)
    elif name == "median_plain_index":  # backward only: the median-depth gradient lands on another entry
        assert same_forward() and _differs(v["A_dL_dmeans3D"], d["A_dL_dmeans3D"], rel=1e-6)
    elif name == "proxy_view_depth":   # scene B carries transMat rows scaled by 2: Tw.z = 2 x view depth -> the shipped proxy is twice this one
        assert same_forward() and np.array_equal(v["A_dL_dmeans3D"], d["A_dL_dmeans3D"])
        big = np.abs(d["B_dL_dmeans2D"]) > 1e-3 * np.abs(d["B_dL_dmeans2D"]).max()
        np.testing.assert_allclose(d["B_dL_dmeans2D"][big], 2.0 * v["B_dL_dmeans2D"][big], rtol=1e-4)
        np.testing.assert_allclose(v["A_dL_dmeans2D"], d["A_dL_dmeans2D"], rtol=1e-4, atol=1e-6 * np.abs(d["A_dL_dmeans2D"]).max())   # reference projection: Tw.z == view z up to rounding
    elif name == "backward_wh_from_size":   # scene C: an image size upstream's int(focal * tanfov * 2) truncates to W - 1 (the SHIPPED build does that)
        assert v["C_found"] and d["C_found"], "no image size in the searched range truncates: widen the search in switch_worker.py"
        Wc = int(v["C_size"][0])
        big = np.abs(v["C_dL_dmeans2D"][:, 0]) > 1e-3 * np.abs(v["C_dL_dmeans2D"][:, 0]).max()
        np.testing.assert_allclose(d["C_dL_dmeans2D"][big, 0] / v["C_dL_dmeans2D"][big, 0], (Wc - 1) / Wc, rtol=1e-5)   # the proxy's W / 2 factor
        assert _differs(v["C_dL_dmeans3D"], d["C_dL_dmeans3D"], rel=1e-5) and same_forward()
    elif name == "pz_zero_through_filter":  # scene D: at pixel column 16 the big splat's p.z is exactly 0 -- the SHIPPED build skips the pair there, like upstream
        col_d, col_v = d["D_color"][:, :, 16], v["D_color"][:, :, 16]
        # (the shipped build leaves that column to the background and the small splats; the variant blends the big splat through its 2-D footprint)
        assert _differs(col_v, col_d, rel=1e-2), "the p.z == 0 column renders the same with and without the per-pair skip"
        others = [x for x in range(33) if x != 16]
        assert np.array_equal(v["D_color"][:, :, others], d["D_color"][:, :, others]) and same_forward()
    else:
        raise AssertionError(name)
