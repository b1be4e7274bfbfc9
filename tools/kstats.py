"""Prints the per-kernel averages of a rocprofv3 --stats CSV (default: the last tools/quick_trace.sh run)."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/quick_*/trace_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "at::" in n: continue
    print(f"{n.split('(')[0].replace('void ', '')[-52:]:54s} x{r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us")
