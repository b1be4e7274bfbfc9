"""Builds streetunveiler_amd/lib/libsurfel_raster.so (HIP kernels + C-ABI) for gfx950 with hipcc.

In-tree build, no JIT cache: the .so is git-ignored but travels with the gpurun snapshot.
    python -m streetunveiler_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsurfel_raster.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -fno-slp-vectorize: packed FP32 (v_pk_fma_f32 ...) is no faster than two scalar ops on gfx950 and costs register-pair moves and
# registers (K7: 228 -> 196 VGPRs, K6: 102 -> 88) -- measured: K6 -14 %, K7 -7 %.
# -disable-promote-alloca-to-vector: keeps the per-lane accumulator arrays as scalars (SROA) instead of 32-register tuples that
# are copied wholesale at every branch join (K7: 62 -> 10 v_mov_b64 in the loop body)
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
          "-mllvm", "-disable-promote-alloca-to-vector", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("SR_EXTRA_HIPCC_FLAGS", "").split()   # A/B experiments (-DSR_...=...); empty for the shipped build
# (source, extra flags)
SOURCES = [
    ("preprocess.hip", ["-ffp-contract=off"]),  # op order is part of the bit-exact contract with the oracle
    ("binning.hip", []),
    ("radix_sort.hip", []),
    ("render.hip", []),
    ("render_bwd.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),   # K7: -1 % with the max-ILP scheduler (the forward kernel spills with it)
    ("render_class.hip", []),
    ("postprocess.hip", []),
    ("knn.hip", ["-ffp-contract=off"]),         # squared distances bit-identical to the brute-force oracle
    ("api.hip", []),
]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "blend_common.h"), os.path.join(os.path.dirname(HERE), "include", "surfel_raster.h"),
           os.path.join(os.path.dirname(HERE), "include", "surfel_switches.h")]

# The named switches of include/surfel_switches.h at their non-default values (SURVEY.md Appendix A's (!) items): each is a complete
# build of kernels AND oracle with the same -D, in lib/variants/<name>/ (libsurfel_raster.so + libsurfel_oracle.so); a maintainer
# holding the real CUDA fork ships the one that matches it (SURFEL_RASTER_LIB=...), tests/test_gpu_switches.py checks every one of them.
VARIANTS = {
    "tightbbox": ["-DSR_TIGHTBBOX=1"],
    "detach_weight": ["-DSR_DETACH_WEIGHT=1"],
    "no_radius_floor": ["-DSR_RADIUS_FILTER_FLOOR=0"],
    "median_plain_index": ["-DSR_MEDIAN_CONTRIBUTOR_MINUS_ONE=0"],
    "proxy_view_depth": ["-DSR_PROXY_DEPTH_VIEW_Z=1"],
    "backward_wh_from_focal": ["-DSR_BACKWARD_WH_FROM_FOCAL=1"],
    "reference_pz_skip": ["-DSR_REFERENCE_PZ_SKIP=1"],
}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def variant_dir(name: str) -> str:
    return os.path.join(LIBDIR, "variants", name)


def build_variant(name: str, force: bool = False, verbose: bool = False):
    """Kernels + oracle with one named switch flipped -> (libsurfel_raster.so, libsurfel_oracle.so) in lib/variants/<name>/."""
    defines = VARIANTS[name]
    d = variant_dir(name)
    lib = build(force, verbose, libdir=d, defines=defines)
    oracle_dir = os.path.join(os.path.dirname(HERE), "oracle")
    out = os.path.join(d, "libsurfel_oracle.so")
    srcs = [os.path.join(oracle_dir, f) for f in ("surfel_oracle.c", "surfel_blend.inc", "surfel_k8.inc", "knn_oracle.c")] + [HEADERS[-1]]
    if force or _stale(out, srcs):
        r = subprocess.run(["make", "-C", oracle_dir, "-B", "OUT=" + out, "CFLAGS_EXTRA=" + " ".join(defines)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle variant build failed:\n" + r.stdout + r.stderr)
    return lib, out


def build(force: bool = False, verbose: bool = False, libdir: str = LIBDIR, defines=()) -> str:
    os.makedirs(libdir, exist_ok=True)
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(libdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s, __file__] + HEADERS):
            jobs.append([HIPCC] + COMMON + list(defines) + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(libdir, "libsurfel_raster.so")
    if force or jobs or _stale(lib, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        which = sys.argv[sys.argv.index("--variant") + 1]
        for name in (sorted(VARIANTS) if which == "all" else [which]):
            print(name, *build_variant(name, force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
