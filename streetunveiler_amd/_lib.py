"""ctypes binding of include/surfel_raster.h (the C-ABI of the HIP library).

Fails loudly when the library is missing: there is NO CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SURFEL_RASTER_LIB: another BUILD of the same library -- one of the named-switch variants of include/surfel_switches.h
# (`python -m streetunveiler_amd.build --variant <name>` -> lib/variants/<name>/libsurfel_raster.so).  Same ABI, still no CPU path.
LIB_PATH = os.environ.get("SURFEL_RASTER_LIB") or os.path.join(HERE, "lib", "libsurfel_raster.so")

SR_ACT_EXP_SCALES, SR_ACT_SIGMOID_OPACITY, SR_ACT_NORMALIZE_ROTATIONS = 1, 2, 4
SR_FLAG_NO_QUADRANT_CULL = 1
SR_FLAG_BALLOT_RANKING = 2
SR_FLAG_ROW_MAPPED_FORWARD = 4
SR_FLAG_QUADRANT_MAPPED_FORWARD = 8
SR_FLAG_FORWARD_ONLY = 16
SR_FLAG_NO_PRECOMP_COLOR_GRAD = 32
SR_FLAG_BINNING_CAPACITY = 64
SR_FLAG_ONE_SWEEP_SORT = 128
SR_FLAG_ONE_WAVE_BACKWARD = 256
SR_FLAG_COOP_BACKWARD = 512
SR_FLAG_ROW_BACKWARD = 1024
SR_ABI_VERSION = 10
SR_STAGE_NAMES = ["preprocess", "depth_sort", "scan", "expand_x", "expand_y", "ranges", "blend_fwd", "blend_bwd",
                  "preprocess_bwd", "class_partition", "class_fwd", "class_bwd"]


class SrFrame(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("tile_width", C.c_int32), ("tile_height", C.c_int32), ("flags", C.c_uint32), ("blend_counters", C.c_void_p)]


class SrGaussians(C.Structure):
    _fields_ = [("P", C.c_int32), ("sh_coeffs", C.c_int32), ("color_channels", C.c_int32), ("activations", C.c_int32),
                ("means3D", C.c_void_p), ("opacities", C.c_void_p),
                ("scales", C.c_void_p), ("rotations", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("transMat_precomp", C.c_void_p), ("mask", C.c_void_p)]


class SrGradients(C.Structure):
    _fields_ = [("dL_dmeans2D", C.c_void_p), ("dL_dcolors", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dmeans3D", C.c_void_p), ("dL_dtransMat", C.c_void_p), ("dL_dsh", C.c_void_p),
                ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p)]


class SrGeomView(C.Structure):
    _fields_ = [("splats", C.c_void_p), ("depth_keys", C.c_void_p), ("tiles_touched", C.c_void_p), ("clamped", C.c_void_p),
                ("sorted_gid", C.c_void_p), ("frame_counts", C.c_void_p)]


class SrBinningView(C.Structure):
    _fields_ = [("point_list", C.c_void_p), ("ranges", C.c_void_p), ("tile_order", C.c_void_p)]


class SrImageView(C.Structure):
    _fields_ = [("final_T", C.c_void_p), ("n_contrib", C.c_void_p)]


# every symbol include/surfel_raster.h declares (checked by tests/test_abi.py)
EXPORTS = ["sr_abi_version", "sr_build_switches", "sr_source_digest", "sr_last_error", "sr_geom_bytes", "sr_binning_bytes", "sr_image_bytes",
           "sr_backward_workspace_bytes", "sr_geom_view", "sr_binning_view", "sr_image_view", "sr_forward_plan", "sr_sh_gradient_expand", "sr_knn_workspace_bytes", "sr_knn_mean_dist2",
           "sr_forward_render", "sr_backward", "sr_backward_blend", "sr_backward_colors", "sr_backward_geometry", "sr_debug_pair_decisions", "sr_class_image_bytes", "sr_class_forward_render", "sr_class_backward", "sr_class_shared_bytes", "sr_class_forward_shared", "sr_class_backward_shared", "sr_mark_visible", "sr_set_stage_timing", "sr_stage_stats", "sr_debug_radix_sort", "sr_debug_radix_sort_temp_bytes", "sr_debug_lds_atomic_ranks", "sr_rank_mode", "sr_postprocess_forward",
           "sr_postprocess_backward"]

_lib = None


class SurfelRasterError(RuntimeError):
    pass


def load():
    """Load libsurfel_raster.so; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SurfelRasterError(
            f"{LIB_PATH} is missing: build it with `python -m streetunveiler_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback for the rasterizer.")
    lib = C.CDLL(LIB_PATH)
    # the in-tree library must be the one built from the sources next to it (it travels prebuilt to the GPU box); SURFEL_RASTER_LIB -- a
    # variant or A/B build chosen on purpose -- is taken as it is
    if not os.environ.get("SURFEL_RASTER_LIB"):
        from .build import source_digest
        lib.sr_source_digest.restype = C.c_char_p
        have, want = lib.sr_source_digest().decode(), source_digest()
        if have != want:
            raise SurfelRasterError(f"{LIB_PATH} was built from other sources (digest {have}, the tree's is {want}): rebuild it with "
                                    "`python -m streetunveiler_amd.build`")
    lib.sr_abi_version.restype = C.c_int
    lib.sr_build_switches.restype = C.c_uint32
    lib.sr_last_error.restype = C.c_char_p
    for name in ("sr_geom_bytes", "sr_binning_bytes", "sr_image_bytes", "sr_backward_workspace_bytes"):
        getattr(lib, name).restype = C.c_size_t
    lib.sr_geom_bytes.argtypes = [C.c_int32]
    lib.sr_binning_bytes.argtypes = [C.c_int32, C.c_uint32, C.c_int32, C.c_int32]
    lib.sr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.sr_backward_workspace_bytes.argtypes = [C.c_int32, C.c_uint32, C.c_int32]
    lib.sr_geom_view.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(SrGeomView)]
    lib.sr_binning_view.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(SrBinningView)]
    lib.sr_image_view.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(SrImageView)]
    lib.sr_forward_plan.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.POINTER(C.c_uint32), C.c_void_p]
    lib.sr_forward_render.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sr_backward.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_size_t, C.POINTER(SrGradients), C.c_void_p]
    lib.sr_backward_blend.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sr_backward_colors.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_void_p]
    lib.sr_backward_geometry.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(SrGradients), C.c_void_p]
    lib.sr_debug_pair_decisions.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sr_class_shared_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    lib.sr_class_shared_bytes.restype = C.c_size_t
    lib.sr_class_forward_shared.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.sr_class_backward_shared.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sr_class_image_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.sr_class_image_bytes.restype = C.c_size_t
    lib.sr_class_forward_render.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.sr_class_backward.argtypes = [C.POINTER(SrFrame), C.POINTER(SrGaussians), C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.POINTER(SrGradients), C.c_void_p]
    lib.sr_sh_gradient_expand.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    lib.sr_knn_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.sr_knn_workspace_bytes.restype = C.c_size_t
    lib.sr_knn_mean_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]
    lib.sr_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sr_set_stage_timing.argtypes = [C.c_int]
    lib.sr_postprocess_forward.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 7
    lib.sr_postprocess_backward.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 9
    lib.sr_debug_radix_sort_temp_bytes.argtypes = [C.c_uint32]
    lib.sr_debug_radix_sort_temp_bytes.restype = C.c_size_t
    lib.sr_debug_radix_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    lib.sr_rank_mode.argtypes = [C.c_void_p]
    lib.sr_debug_lds_atomic_ranks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.sr_stage_stats.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    if lib.sr_abi_version() != SR_ABI_VERSION:
        raise SurfelRasterError(f"ABI version mismatch: library reports {lib.sr_abi_version()}, binding expects {SR_ABI_VERSION}")
    _lib = lib
    return lib


# include/surfel_switches.h: bit -> the switch that is NOT at its default in a library reporting that bit (sr_build_switches)
SWITCH_BITS = {1: "SR_TIGHTBBOX=1", 2: "SR_DETACH_WEIGHT=1", 4: "SR_RADIUS_FILTER_FLOOR=0", 8: "SR_MEDIAN_CONTRIBUTOR_MINUS_ONE=0",
               16: "SR_PROXY_DEPTH_VIEW_Z=1", 32: "SR_BACKWARD_WH_FROM_FOCAL=0", 64: "SR_REFERENCE_PZ_SKIP=0"}


def build_switches():
    """Non-default named switches of the loaded library, e.g. ['SR_DETACH_WEIGHT=1']; [] for the shipped configuration (= upstream's
    semantics as SURVEY.md Appendix A states them)."""
    bits = int(load().sr_build_switches())
    return [name for bit, name in SWITCH_BITS.items() if bits & bit]


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sr_last_error().decode("utf-8", "replace")
        raise SurfelRasterError(f"{what} failed ({rc}): {msg}")


def stage_stats():
    """{stage name: (total_ms, launches)} since sr_set_stage_timing(1)."""
    lib = load()
    out = {}
    for i, name in enumerate(SR_STAGE_NAMES):
        ms, n = C.c_float(0), C.c_int(0)
        check(lib.sr_stage_stats(i, C.byref(ms), C.byref(n)), "sr_stage_stats")
        out[name] = (float(ms.value), int(n.value))
    return out
