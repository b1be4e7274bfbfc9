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
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "blend_common.h"), os.path.join(os.path.dirname(HERE), "include", "surfel_raster.h")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s, __file__] + HEADERS):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
