#!/usr/bin/env python
"""The LOCAL kernels of the factored multi-GPU exchange at C3, timed on one GPU: sr_sh_gradient_expand for 1 / 2 / 4 / 8 gathered views
(what every rank runs after the all-gather of the 12-B colour gradients) -- the part of the predicted N-GPU step that needs no link.
    python tools/time_exchange_kernels.py [gaussians]   (GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import _C
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
means = (torch.randn(P, 3, generator=g) * 10).to(dev)
out = {"gaussians": P}
for V in (1, 2, 4, 8):
    campos = torch.randn(V, 3, generator=g).to(dev); gc = torch.randn(V, P, 3, generator=g).to(dev)
    for _ in range(3): _C.sh_gradient_expand(means, campos, gc, 16, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): _C.sh_gradient_expand(means, campos, gc, 16, 3)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    nbytes = P * (12 + V * 12 + 192)
    out[f"sh_gradient_expand_{V}_views"] = {"ms": round(ms, 4), "bytes": nbytes, "GBs": round(nbytes / ms / 1e6, 1)}
    del gc
print(json.dumps(out))
