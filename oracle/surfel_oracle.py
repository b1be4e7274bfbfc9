"""ctypes front-end for oracle/surfel_oracle.c (CPU restatement of the surfel rasterizer).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from the product packages
(streetunveiler_amd/, diff_surfel_rasterization/).

Parity status: "parity unpinned" for the rasterizer as a whole (see the header
of surfel_oracle.c); SH / camera matrices / quaternion pieces are pinned by
tests/golden/ against the importable reference python.

The argument names and tensor layouts mirror the operator boundary
[REF /root/reference/gaussian_renderer/__init__.py:39-52,129-138].
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SURFEL_ORACLE_LIB: the oracle compiled with a non-default named switch (include/surfel_switches.h; build_variant() below)
_LIB_PATH = os.environ.get("SURFEL_ORACLE_LIB") or os.path.join(_HERE, "libsurfel_oracle.so")
_lib = None

TILE = 16


_SOURCES = ("surfel_oracle.c", "surfel_blend.inc", "surfel_k8.inc", "knn_oracle.c", "../include/surfel_switches.h")


def _make(out: str, defines=(), force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in _SOURCES + ("Makefile",)]
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        r = subprocess.run(["make", "-C", _HERE, "-B", "OUT=" + out, "CFLAGS_EXTRA=" + " ".join(defines)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return out


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    if os.environ.get("SURFEL_ORACLE_LIB"):
        return _LIB_PATH   # a variant: built by build_variant()
    return _make(_LIB_PATH, force=force)


def variant_path(name: str) -> str:
    return os.path.join(_HERE, "variants", name, "libsurfel_oracle.so")


def build_variant(name: str, force: bool = False) -> str:
    """The oracle with one named switch of include/surfel_switches.h flipped (the same -D as the kernels' variant of that name:
    streetunveiler_amd.build.VARIANTS) -> oracle/variants/<name>/libsurfel_oracle.so.  Selected with SURFEL_ORACLE_LIB=<path>."""
    from streetunveiler_amd.build import VARIANTS
    return _make(variant_path(name), VARIANTS[name], force)


def use_native_build() -> bool:
    """bench.py's cpu_baseline leg: switch this process to oracle/_native/libsurfel_oracle.so, compiled here and now with -O3 -march=native
    (make native: BASELINE.md 3's flags; the default build must stay portable because it travels to the GPU box as a file).  Call before the
    first use of the library.  False (and the portable build stays) if it cannot be compiled on this machine."""
    global _LIB_PATH, _lib
    assert _lib is None, "use_native_build() must come before the first oracle call of the process"
    r = subprocess.run(["make", "-C", _HERE, "-B", "native"], capture_output=True, text=True)
    out = os.path.join(_HERE, "_native", "libsurfel_oracle.so")
    if r.returncode != 0 or not os.path.exists(out):
        return False
    _LIB_PATH = out
    return True


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.so_count_duplicates.restype = C.c_uint64
        _lib.so_bin.restype = C.c_int
        _lib.so_num_threads.restype = C.c_int
        _lib.so_build_switches.restype = C.c_uint32
        _lib.so_set_tanfov.argtypes = [C.c_float, C.c_float]
    return _lib


def build_switches() -> int:
    """SR_SWITCH_BITS the oracle was compiled with (include/surfel_switches.h); 0 = the shipped configuration."""
    return int(lib().so_build_switches())


def quat_to_R(q) -> np.ndarray:
    """The oracle's quaternion (r, x, y, z) -> rotation restatement, [n,4] -> [n,3,3] (used as given, no normalisation)."""
    q = _f32(q); out = np.zeros((q.shape[0], 3, 3), np.float32)
    lib().so_quat_to_R(q.shape[0], _p(q), _p(out))
    return out


def num_threads() -> int:
    return int(lib().so_num_threads())


def _f32(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, ty=C.c_float):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ty))


_GEOMETRY_KEYS = ("radii", "means2D", "depths", "transMat", "normal_opacity", "rgb", "clamped", "tiles_touched", "rect",
                  "num_rendered", "keys", "point_list", "ranges")


def _preprocess_and_bin(L, P, deg, M, means3D, scales, rotations, opacities, shs, colors_precomp, transMat_precomp, view, proj, cam,
                        W, H, scale_modifier, tile):
    """K1 (so_preprocess_forward) and K2-K5 (so_bin: duplicate keys, stable 64-bit sort, tile ranges)."""
    o = dict(
        radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
        transMat=np.zeros((P, 9), np.float32), normal_opacity=np.zeros((P, 4), np.float32),
        rgb=np.zeros((P, 3), np.float32), clamped=np.zeros((P, 3), np.uint8),
        tiles_touched=np.zeros(P, np.uint32), rect=np.zeros((P, 4), np.int32))
    L.so_preprocess_forward(P, deg, M, _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(shs),
                            _p(colors_precomp), _p(transMat_precomp), _p(view), _p(proj), _p(cam), W, H,
                            C.c_float(scale_modifier), _p(o["radii"], C.c_int32), _p(o["means2D"]), _p(o["depths"]),
                            _p(o["transMat"]), _p(o["normal_opacity"]), _p(o["rgb"]), _p(o["clamped"], C.c_uint8),
                            _p(o["tiles_touched"], C.c_uint32), _p(o["rect"], C.c_int32))
    D = int(L.so_count_duplicates(P, _p(o["tiles_touched"], C.c_uint32)))
    gx, gy = (W + int(tile[0]) - 1) // int(tile[0]), (H + int(tile[1]) - 1) // int(tile[1])
    o["num_rendered"] = D
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    keys_buf = np.zeros(max(D, 1), np.uint64); vals_buf = np.zeros(max(D, 1), np.uint32)
    rc = L.so_bin(P, W, H, _p(o["radii"], C.c_int32), _p(o["depths"]), _p(o["rect"], C.c_int32),
                  _p(o["tiles_touched"], C.c_uint32), C.c_uint64(D), _p(keys_buf, C.c_uint64),
                  _p(vals_buf, C.c_uint32), _p(o["ranges"], C.c_uint32))
    assert rc == 0, f"so_bin failed: {rc}"
    o["keys"] = keys_buf[:D]; o["point_list"] = vals_buf[:D]
    return o, vals_buf


def rasterize_forward(means3D, opacities, scales=None, rotations=None, shs=None, colors_precomp=None,
                      transMat_precomp=None, *, viewmatrix, projmatrix, campos, bg, image_width: int,
                      image_height: int, sh_degree: int = 0, scale_modifier: float = 1.0,
                      stages: bool = True, tile=(16, 16), forced=None, f64: bool = False, reuse=None,
                      tanfovx: Optional[float] = None, tanfovy: Optional[float] = None) -> Dict[str, np.ndarray]:
    """K1..K6. Returns every stage's outputs (dict of numpy arrays).  `tile` = (BLOCK_X, BLOCK_Y), 16x16 in the reference.
    `forced` = dict(valid=u64[D,nq], use3d=u64[D,nq], n_contrib=u32[2,H,W]): blend with the hard decisions of another
    implementation (sr_debug_pair_decisions + its n_contrib) -- see so_render_forward; the backward then uses them too.
    `f64=True`: the blend (K6, and K7 in rasterize_backward) is evaluated in double precision on the same float32 per-Gaussian
    inputs (surfel_blend.inc compiled with REAL = double): the arbiter of the parity report, not the oracle.
    `reuse` = the dict of an earlier call on the SAME inputs: K1 and the binning (the 64-bit key sort) are taken from it and only the
    blend runs again (forced / float64 variants of one scene at full size)."""
    L = lib()
    L.so_set_tile(int(tile[0]), int(tile[1]))
    means3D = _f32(means3D); P = means3D.shape[0]
    opacities = _f32(opacities).reshape(P)
    scales = _f32(scales); rotations = _f32(rotations); shs = _f32(shs)
    colors_precomp = _f32(colors_precomp); transMat_precomp = _f32(transMat_precomp)
    assert (shs is None) != (colors_precomp is None), "exactly one of shs / colors_precomp"
    assert (transMat_precomp is None) != (scales is None or rotations is None), "exactly one of (scales,rotations) / transMat_precomp"
    view = _f32(viewmatrix).reshape(16); proj = _f32(projmatrix).reshape(16)
    cam = _f32(campos).reshape(3); bgc = _f32(bg).reshape(3)
    W, H = int(image_width), int(image_height)
    M = shs.shape[1] if shs is not None else 0
    if reuse is not None:   # K1 + binning of an earlier call on the same inputs
        assert reuse["_inputs"]["tile"] == (int(tile[0]), int(tile[1])) and reuse["radii"].shape[0] == P
        o = {k: reuse[k] for k in _GEOMETRY_KEYS}
        vals_buf = reuse["_inputs"]["vals_buf"]
    else:
        o, vals_buf = _preprocess_and_bin(L, P, int(sh_degree), M, means3D, scales, rotations, opacities, shs, colors_precomp,
                                          transMat_precomp, view, proj, cam, W, H, float(scale_modifier), tile)
    D = int(o["num_rendered"])
    rt, ct = (np.float64, C.c_double) if f64 else (np.float32, C.c_float)
    o["color"] = np.zeros((3, H, W), rt); o["allmap"] = np.zeros((7, H, W), rt)
    o["final_T"] = np.zeros((3, H, W), rt); o["n_contrib"] = np.zeros((2, H, W), np.uint32)
    tested = C.c_uint64(0)
    fv = fu = fn = None
    if forced is not None:
        nq = (int(tile[0]) // 8) * (int(tile[1]) // 8)
        fv = np.ascontiguousarray(forced["valid"], dtype=np.uint64).reshape(-1); fu = np.ascontiguousarray(forced["use3d"], dtype=np.uint64).reshape(-1)
        fn = np.ascontiguousarray(forced["n_contrib"], dtype=np.uint32).reshape(2, H, W)
        assert fv.size == D * nq and fu.size == D * nq, "forced decisions must cover every (list position, quadrant)"
    (L.so_render_forward_f64 if f64 else L.so_render_forward)(
        W, H, _p(o["ranges"], C.c_uint32), _p(vals_buf, C.c_uint32), _p(o["means2D"]), _p(o["transMat"]), _p(o["normal_opacity"]),
        _p(o["rgb"]), _p(bgc), _p(o["color"], ct), _p(o["allmap"], ct), _p(o["final_T"], ct), _p(o["n_contrib"], C.c_uint32),
        C.byref(tested), _p(fv, C.c_uint64), _p(fu, C.c_uint64), _p(fn, C.c_uint32))
    o["tested_pairs"] = int(tested.value)
    o["_inputs"] = dict(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations, shs=shs,
                        colors_precomp=colors_precomp, transMat_precomp=transMat_precomp, view=view, proj=proj,
                        cam=cam, bg=bgc, W=W, H=H, deg=int(sh_degree), M=M, scale_modifier=float(scale_modifier),
                        vals_buf=vals_buf, tile=(int(tile[0]), int(tile[1])), forced=(fv, fu), f64=bool(f64),
                        tanfov=(tanfovx, tanfovy))   # K8's image size (SR_BACKWARD_WH_FROM_FOCAL, the default)
    return o


def rasterize_backward(fwd: Dict[str, np.ndarray], dL_dcolor, dL_dallmap) -> Dict[str, np.ndarray]:
    """K7 + K8 on the state returned by rasterize_forward."""
    L = lib()
    i = fwd["_inputs"]; P = i["means3D"].shape[0]; W, H, M = i["W"], i["H"], i["M"]
    L.so_set_tile(*i["tile"])
    if i.get("tanfov", (None, None))[0] is not None:
        L.so_set_tanfov(float(i["tanfov"][0]), float(i["tanfov"][1]))
    elif not (build_switches() & 32):
        raise ValueError("upstream's backward derives the image size from int(focal * tanfov * 2) (SR_BACKWARD_WH_FROM_FOCAL=1, the default): "
                         "pass the operator's tanfovx / tanfovy to rasterize_forward")
    dL_dcolor = _f32(dL_dcolor).reshape(3, H, W); dL_dallmap = _f32(dL_dallmap).reshape(7, H, W)
    f64 = i.get("f64", False)
    rt, ct = (np.float64, C.c_double) if f64 else (np.float32, C.c_float)
    g = dict(dL_dcolors=np.zeros((P, 3), rt), dL_dnormal3D=np.zeros((P, 3), rt), dL_dtransMat=np.zeros((P, 9), rt),
             dL_dmean2D_raw=np.zeros((P, 2), rt), dL_dopacity=np.zeros((P, 1), rt))
    (L.so_render_backward_f64 if f64 else L.so_render_backward)(
        W, H, _p(fwd["ranges"], C.c_uint32), _p(i["vals_buf"], C.c_uint32), _p(fwd["means2D"]), _p(fwd["transMat"]),
        _p(fwd["normal_opacity"]), _p(fwd["rgb"]), _p(i["bg"]), _p(fwd["final_T"], ct), _p(fwd["n_contrib"], C.c_uint32),
        _p(dL_dcolor), _p(dL_dallmap), _p(g["dL_dcolors"], ct), _p(g["dL_dnormal3D"], ct), _p(g["dL_dtransMat"], ct),
        _p(g["dL_dmean2D_raw"], ct), _p(g["dL_dopacity"], ct), _p(i["forced"][0], C.c_uint64), _p(i["forced"][1], C.c_uint64))
    if f64:
        # the float64 arbiter of the WHOLE backward: K8 in double precision on the double sums (surfel_k8.inc, REAL = double) ->
        # "<name>64"; then, as before, the float32 K8 on the double sums rounded once -> "<name>"
        g["dL_dopacity64"] = g["dL_dopacity"].copy()
        g64 = dict(dL_dmeans3D64=np.zeros((P, 3)), dL_dscales64=np.zeros((P, 2)), dL_drotations64=np.zeros((P, 4)), dL_dmeans2D64=np.zeros((P, 3)))
        dsh64 = np.zeros((P, max(M, 1), 3))
        dT64 = g["dL_dtransMat"].copy()
        L.so_preprocess_backward_f64(P, i["deg"], M, _p(i["means3D"]), _p(i["scales"]), _p(i["rotations"]), _p(i["shs"]),
                                     _p(i["transMat_precomp"]), _p(i["view"]), _p(i["proj"]), _p(i["cam"]), W, H,
                                     C.c_double(i["scale_modifier"]), _p(fwd["radii"], C.c_int32), _p(fwd["clamped"], C.c_uint8),
                                     _p(fwd["transMat"]), _p(dT64, C.c_double), _p(g["dL_dnormal3D"], C.c_double), _p(g["dL_dmean2D_raw"], C.c_double),
                                     _p(g["dL_dcolors"], C.c_double), _p(g64["dL_dmeans3D64"], C.c_double), _p(g64["dL_dscales64"], C.c_double),
                                     _p(g64["dL_drotations64"], C.c_double), _p(dsh64, C.c_double), _p(g64["dL_dmeans2D64"], C.c_double))
        g64["dL_dsh64"] = dsh64 if M else dsh64[:, :0]
        g64["dL_dtransMat64"] = dT64
        for k in ("dL_dcolors", "dL_dnormal3D", "dL_dtransMat", "dL_dmean2D_raw", "dL_dopacity"):
            g[k] = np.ascontiguousarray(g[k], dtype=np.float32)
        g.update(g64)
    g["dL_dtransMat_render"] = g["dL_dtransMat"].copy()
    g["dL_dmeans3D"] = np.zeros((P, 3), np.float32); g["dL_dscales"] = np.zeros((P, 2), np.float32)
    g["dL_drotations"] = np.zeros((P, 4), np.float32); g["dL_dmeans2D"] = np.zeros((P, 3), np.float32)
    g["dL_dsh"] = np.zeros((P, max(M, 1), 3), np.float32)[:, :M]
    dsh_buf = np.zeros((P, max(M, 1), 3), np.float32)
    L.so_preprocess_backward(P, i["deg"], M, _p(i["means3D"]), _p(i["scales"]), _p(i["rotations"]), _p(i["shs"]),
                             _p(i["transMat_precomp"]), _p(i["view"]), _p(i["proj"]), _p(i["cam"]), W, H,
                             C.c_float(i["scale_modifier"]), _p(fwd["radii"], C.c_int32), _p(fwd["clamped"], C.c_uint8),
                             _p(fwd["transMat"]), _p(g["dL_dtransMat"]), _p(g["dL_dnormal3D"]), _p(g["dL_dmean2D_raw"]),
                             _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]),
                             _p(dsh_buf), _p(g["dL_dmeans2D"]))
    if M:
        g["dL_dsh"] = dsh_buf
    return g


# noise allowances of the blend's hard decisions (see so_render_margins): float32 rounding of exp / rcp / the ray-splat cross
# product moves 255*alpha by ~1e-6, T' by ~1e-5 after ~100 factors, rho3d (tile-local vs global pixel coordinates) by ~1e-4
DEFAULT_EPS = dict(alpha=2e-5, T=1e-4, path=1e-3, near=1e-5, median=2e-5)


def render_margins(fwd: Dict[str, np.ndarray], eps: Optional[Dict[str, float]] = None, f64: bool = False,
                   kernel_decisions: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """Decision margins of the forward blend (so_render_margins): `pixel`[H,W], `median`[H,W], `gaussian`[P]; > 1 = robust.
    `f64=True`: the deciding quantities are evaluated in double precision on the float32 per-Gaussian state (so_render_margins_f64),
    i.e. the margins are distances of the TRUE decisions from their thresholds, free of this oracle's own float32 noise.
    `kernel_decisions` = dict(valid=u64[D,nq], use3d=u64[D,nq]) (sr_debug_pair_decisions): adds `disagree`[H,W], the number of pairs of
    each pixel -- up to where this walk stops -- whose contribute / path decision differs from the walk's own."""
    L = lib()
    i = fwd["_inputs"]; P = i["means3D"].shape[0]; W, H = i["W"], i["H"]
    L.so_set_tile(*i["tile"])
    e = dict(DEFAULT_EPS); e.update(eps or {})
    ev = np.array([e["alpha"], e["T"], e["path"], e["near"], e["median"]], np.float32)
    pm = np.zeros((H, W), np.float32); mm = np.zeros((H, W), np.float32); gm = np.zeros(P, np.float32); vn = np.zeros((H, W), np.float32)
    kv = ku = dis = None
    if kernel_decisions is not None:
        nq = (i["tile"][0] // 8) * (i["tile"][1] // 8); D = int(fwd["num_rendered"])
        kv = np.ascontiguousarray(kernel_decisions["valid"], dtype=np.uint64).reshape(-1); ku = np.ascontiguousarray(kernel_decisions["use3d"], dtype=np.uint64).reshape(-1)
        assert kv.size == D * nq and ku.size == D * nq, "kernel decisions must cover every (list position, quadrant)"
        dis = np.zeros((H, W), np.uint32)
    (L.so_render_margins_f64 if f64 else L.so_render_margins)(P, W, H, _p(fwd["ranges"], C.c_uint32), _p(i["vals_buf"], C.c_uint32), _p(fwd["means2D"]),
                        _p(fwd["transMat"]), _p(fwd["normal_opacity"]), _p(ev), _p(pm), _p(mm), _p(gm), _p(vn), _p(kv, C.c_uint64), _p(ku, C.c_uint64),
                        _p(dis, C.c_uint32))
    # `value_noise`[H,W]: relative noise float32 rounding of the ray-splat intersections can put into the pixel's transmittance and weights
    out = dict(pixel=pm, median=mm, gaussian=gm, value_noise=vn, eps=e)
    if dis is not None:
        out["disagree"] = dis
    return out


def pz_zero_census(fwd: Dict[str, np.ndarray], n_contrib=None) -> Dict[str, int]:
    """Upstream's per-pair `if (p.z == 0) continue` on a forward's lists (so_pz_zero_census): how many (pixel, splat) pairs have a float32
    (k x l).z of exactly 0 up to each pixel's last contributor, how many of them would be visible through their 2-D filter footprint, the
    pixels holding one, and the pairs of identically-degenerate splats.  `n_contrib` = another implementation's [2,H,W] (default: this forward's)."""
    i = fwd["_inputs"]; W, H = i["W"], i["H"]
    L = lib(); L.so_set_tile(*i["tile"])
    nc = np.ascontiguousarray(fwd["n_contrib"] if n_contrib is None else n_contrib, dtype=np.uint32).reshape(2, H, W)
    out = np.zeros(4, np.uint64)
    L.so_pz_zero_census(W, H, _p(fwd["ranges"], C.c_uint32), _p(i["vals_buf"], C.c_uint32), _p(fwd["means2D"]), _p(fwd["transMat"]),
                        _p(fwd["normal_opacity"]), _p(nc, C.c_uint32), _p(out, C.c_uint64))
    return dict(pairs_with_pz_exactly_zero=int(out[0]), of_them_visible_through_the_2d_footprint=int(out[1]), pixels_holding_one=int(out[2]),
                pairs_of_identically_degenerate_splats=int(out[3]))


def mark_visible(means3D, viewmatrix) -> np.ndarray:
    means3D = _f32(means3D); P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().so_mark_visible(P, _p(means3D), _p(_f32(viewmatrix).reshape(16)), _p(out, C.c_uint8))
    return out.astype(bool)
