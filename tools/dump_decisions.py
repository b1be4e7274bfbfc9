#!/usr/bin/env python
"""One forward of the synthetic benchmark scene through whatever kernel library this process loads (SURFEL_RASTER_LIB selects a
named-switch variant), leaving colour, allmap, n_contrib and the per-pair decision ballots in an .npz -- tools/parity_report.py runs
it under the pz_zero_through_filter variant to count the pairs upstream's `if (p.z == 0) continue` removes in the shipped kernels.

    SURFEL_RASTER_LIB=streetunveiler_amd/lib/variants/pz_zero_through_filter/libsurfel_raster.so python tools/dump_decisions.py P W H out.npz
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if __name__ == "__main__":
    P, W, H, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from streetunveiler_amd import _lib
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
    from tests.gpu_util import run_hip_raw
    raw = run_hip_raw(synthetic_gaussians(P, W, H, seed=0), synthetic_camera(W, H), np.zeros(3, np.float32), 3, decisions=True)
    np.savez(out, switches=np.array(_lib.load().sr_build_switches()), valid=raw["decisions"]["valid"], use3d=raw["decisions"]["use3d"],
             color=raw["color"], allmap=raw["allmap"], n_contrib=raw["img"]["n_contrib"].view(np.uint32))
