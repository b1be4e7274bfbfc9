import json, sys
sys.path.insert(0, ".")
import tests.test_gpu_fullsize as t
from streetunveiler_amd.synthetic import clustered_gaussians
out = {}
for tag, P, aux, scene in (("C2", 500_000, False, None), ("C3", 3_000_000, True, None), ("clustered", 3_000_000, True, lambda P, W, H: clustered_gaussians(P, W, H, 0.5))):
    rep = {}
    t._against_oracle(P, aux, tag, pixel_budget=2e-2, gaussian_budget=0.5, scene=scene, report=rep)
    out[tag] = {k: v for k, v in rep.items()}
    print(tag, "non-robust px", rep["non_robust_pixels"], "gaussians", rep["non_robust_visible_gaussians"])
    print("  pixel non_robust_max", max(v["non_robust_max"] for k, v in rep.items() if isinstance(v, dict) and "non_robust_max" in v and "allmap[5]" not in k))
    print("  row non_robust_max_of_tensor_scale", max(v["non_robust_max_of_tensor_scale"] for k, v in rep.items() if isinstance(v, dict) and "non_robust_max_of_tensor_scale" in v))
    print("  robust px max", max(v["robust_max"] for k, v in rep.items() if isinstance(v, dict) and "non_robust_max" in v), " robust rows max", max(v["robust_max"] for k, v in rep.items() if isinstance(v, dict) and "robust_rows" in v))
json.dump(out, open("gpurun_out/nonrobust_report.json", "w"), indent=1, default=float)
