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
    ("build_id.hip", []),                       # + -DSR_SOURCE_DIGEST="..." (build()): recompiled whenever anything else is
]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "blend_common.h"), os.path.join(os.path.dirname(HERE), "include", "surfel_raster.h"),
           os.path.join(os.path.dirname(HERE), "include", "surfel_switches.h")]

# The named switches of include/surfel_switches.h at their non-default values (SURVEY.md Appendix A's (!) items): each is a complete
# build of the kernels with that -D in lib/variants/<name>/libsurfel_raster.so; a maintainer holding the real CUDA fork ships the one
# that matches it (SURFEL_RASTER_LIB=...).  The CPU oracle of the same switch is test infrastructure and lives under oracle/variants/
# (oracle.surfel_oracle.build_variant); tests/test_gpu_switches.py builds both on demand and checks every one of them.
VARIANTS = {
    "tightbbox": ["-DSR_TIGHTBBOX=1"],
    "detach_weight": ["-DSR_DETACH_WEIGHT=1"],
    "no_radius_floor": ["-DSR_RADIUS_FILTER_FLOOR=0"],
    "median_plain_index": ["-DSR_MEDIAN_CONTRIBUTOR_MINUS_ONE=0"],
    "proxy_view_depth": ["-DSR_PROXY_DEPTH_VIEW_Z=1"],
    "backward_wh_from_size": ["-DSR_BACKWARD_WH_FROM_FOCAL=0"],      # the exact image size instead of upstream's int(focal * tanfov * 2)
    "pz_zero_through_filter": ["-DSR_REFERENCE_PZ_SKIP=0"],          # the exact-arithmetic rule instead of upstream's per-pair p.z == 0 skip
}


def source_digest() -> str:
    """sha256 over everything that determines the shipped kernels: csrc/*, include/*.h and this script (flags).  tools/collect_profiles.py
    stores it with every rocprofv3 summary; bench.py quotes a committed counter profile only if the digest still matches (there is no
    .git on the GPU box, so the check is on content, not on commits)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + \
        sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")) + [os.path.abspath(__file__)]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def embedded_digest(lib: str):
    """The source digest a built library carries (csrc/build_id.hip), read from the file without loading it; None if it has none."""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return None
    i = data.find(b"SR_SOURCE_DIGEST=")
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + len(b"SR_SOURCE_DIGEST="):j].decode("ascii", "replace")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def variant_dir(name: str) -> str:
    return os.path.join(LIBDIR, "variants", name)


def build_variant(name: str, force: bool = False, verbose: bool = False) -> str:
    """The kernels with one named switch flipped -> lib/variants/<name>/libsurfel_raster.so."""
    return build(force, verbose, libdir=variant_dir(name), defines=VARIANTS[name])


def build(force: bool = False, verbose: bool = False, libdir: str = LIBDIR, defines=()) -> str:
    os.makedirs(libdir, exist_ok=True)
    lib = os.path.join(libdir, "libsurfel_raster.so")
    # A library is current when it CARRIES the digest of the sources next to it (content, not time stamps: a checkout or a copy resets
    # those) -- even where its object files did not travel (the variants' .o files are not shipped to the GPU box).
    digest = source_digest()
    if not force and embedded_digest(lib) == digest:
        return lib
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(libdir, src.replace(".hip", ".o"))
        objs.append(o)
        if src == "build_id.hip":   # always: it is what stamps the library (one second of hipcc)
            jobs.append([HIPCC] + COMMON + list(defines) + extra + ['-DSR_SOURCE_DIGEST="' + digest + '"', "-c", s, "-o", o])
        elif force or _stale(o, [s, __file__] + HEADERS):
            jobs.append([HIPCC] + COMMON + list(defines) + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=max(4, (os.cpu_count() or 4))) as ex:
        list(ex.map(run, jobs))
    run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    if embedded_digest(lib) != digest:
        raise RuntimeError(f"{lib}: built, but it carries digest {embedded_digest(lib)!r} instead of {digest!r}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        which = sys.argv[sys.argv.index("--variant") + 1]
        for name in (sorted(VARIANTS) if which == "all" else [which]):
            print(name, build_variant(name, force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
