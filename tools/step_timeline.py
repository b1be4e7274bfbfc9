#!/usr/bin/env python
"""Timeline of ONE benchmark step from a rocprofv3 kernel trace (tools/quick_trace.sh -> gpurun_out/quick_<cfg>/trace_kernel_trace.csv):
every kernel of the step with its start relative to the step's first kernel, its duration, and the idle gap in front of it -- where
the GPU waits for the host (the num_rendered read-back) or for nothing at all.
    python tools/step_timeline.py [trace_kernel_trace.csv] [step index from the end, default 3]"""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/quick_*/trace_kernel_trace.csv"))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(({"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(f))), key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "preprocess_forward_kernel" in r["name"]]
a, b = starts[-back - 1], starts[-back]
t0 = rows[a]["s"]
prev = None
busy = 0
for r in rows[a:b]:
    gap = 0 if prev is None else r["s"] - prev
    busy += r["e"] - r["s"]
    print(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:8.1f}  gap {gap / 1e3:6.1f}  {r['name'][-60:]}")
    prev = max(prev or 0, r["e"])
print(f"step: {(rows[b]['s'] - t0) / 1e3:.1f} us from K1 to the next K1, kernels busy {busy / 1e3:.1f} us")
