#!/usr/bin/env python
"""Turns the rocprofv3 CSVs of one profiling call (tools/profile_round.sh -> gpurun_out/<dir>) into the evidence files
<tag>_kernel_stats.csv, <tag>_hbm_traffic.json, <tag>_sq_counters.json, <tag>_bench.json (tag = round + workload, e.g. r02_c3).
    python tools/collect_profiles.py gpurun_out/r02_c3 r02_c3 [output dir, default profiles/]
The profiling call itself (on the GPU box):
    rocprofv3 --kernel-trace --stats --output-format csv -d D -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d D -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --kernel-trace --output-format csv -d D -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
"""
import collections, csv, json, os, re, sys

import subprocess
src, tag = sys.argv[1], sys.argv[2]
ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:    # which build the counters describe (run this script in the build container, where .git is; on the GPU box there is none)
    HEAD = subprocess.run(["git", "-C", ROOT_, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    if subprocess.run(["git", "-C", ROOT_, "status", "--porcelain", "--", "streetunveiler_amd/csrc", "bench.py"], capture_output=True, text=True).stdout.strip():
        HEAD += "+uncommitted kernel changes"
except Exception:
    HEAD = os.environ.get("SR_HEAD")
sys.path.insert(0, ROOT_)
from streetunveiler_amd.build import source_digest
# source_digest: content hash of csrc/ + include/ + build.py AS PROFILED -- bench.py refuses to quote these counters once it differs
META = {"round": tag.split("_")[0], "head": HEAD, "source_digest": source_digest()}
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def short(name, pool=True):
    n = name.split("(")[0] if name.startswith(("sr::", "void sr::")) else name[:100]
    n = n.replace("void ", "")
    if "render_forward_kernel<true" in n:
        return "sr::render_forward_kernel[counter variant, 1 untimed launch]"   # bench.py's blend_counts pass (device atomics)
    if pool:   # (K6 is whichever forward blend kernel the frame ran: the per-frame pick, or a forced mapping)
        n = n.replace("render_forward_auto_kernel", "render_forward_kernel").replace("render_forward_rows_kernel", "render_forward_kernel")
    return re.sub(r"<[^<>]*>", "", n) if pool else n   # template arguments dropped: variants of one kernel are pooled


with open(f"{dst}/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in csv.DictReader(open(f"{src}/trace_kernel_stats.csv")):
        w.writerow([short(r["Name"], pool=False), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def per_kernel(fn, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fe, wr = per_kernel(f"{src}/fetch_counter_collection.csv", "FETCH_SIZE"), per_kernel(f"{src}/write_counter_collection.csv", "WRITE_SIZE")
out = {k: {"FETCH_SIZE_KiB": round(fe[k]), "WRITE_SIZE_KiB": round(wr.get(k, 0)), "traffic_bytes_per_launch": int((2 * fe[k] + wr.get(k, 0)) * 1024)}
       for k in fe if k.startswith("sr::")}
json.dump({**META, "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (--kernel-trace only), python bench.py --config <cfg> --steps 3 --warmup 1; "
                  "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (FETCH_SIZE reads 1/2 of a wide coalesced "
                  "stream on gfx950 -- confirmed on preprocess_forward_kernel: 2*FETCH = the 232 B x 3 M it reads; WRITE_SIZE is exact on streaming "
                  "stores; for the 16-B record gathers of the blend kernels the 2x is an upper bound)",
           "kernels": out}, open(f"{dst}/{tag}_hbm_traffic.json", "w"), indent=1)
def durations(fn):
    """average kernel duration (ns) of one profiling pass, from its own kernel trace"""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


passes = {"sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"],
          "sq2": ["SQ_INSTS_VALU_TRANS_F32", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_BRANCH",
                  "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_SCA"],
          "sq3": ["SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_SMEM",
                  "SQ_INSTS_VMEM", "SQ_INSTS_FLAT"],
          "l2": ["TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]}
sq = {}
for run, names in passes.items():
    fn = f"{src}/{run}_counter_collection.csv"
    if os.path.exists(fn):
        for n in names:
            for k, v in per_kernel(fn, n).items():
                if k.startswith("sr::"):
                    sq.setdefault(k, {})[n] = round(v)
# effective shader clock of every kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's duration in the SAME pass -- the chip
# clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"): the VALU-heavy blend kernels run slower than the streaming ones
fn = f"{src}/grbm_counter_collection.csv"
if os.path.exists(fn):
    act, dur = per_kernel(fn, "GRBM_GUI_ACTIVE"), durations(f"{src}/grbm_kernel_trace.csv")
    for k, v in act.items():
        if k.startswith("sr::") and dur.get(k):
            sq.setdefault(k, {})["GRBM_GUI_ACTIVE"] = round(v)
            sq[k]["clock_ghz"] = round(v / 8.0 / dur[k], 4)
            sq[k]["duration_ns_in_clock_pass"] = round(dur[k])
if sq:
    json.dump({**META, "how": "rocprofv3 --pmc <8 SQ counters> --kernel-trace in separate passes (" + " | ".join(" ".join(v) for v in passes.values()) +
                      " | GRBM_GUI_ACTIVE), bench.py --config <cfg> --steps 3 --warmup 1; averages per launch; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles "
                      "(one issue slot of a SIMD); clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / the kernel's duration in that pass",
               "kernels": sq}, open(f"{dst}/{tag}_sq_counters.json", "w"), indent=1)
if os.path.exists(f"{src}/bench.json") and os.path.getsize(f"{src}/bench.json") > 10:
    json.dump(json.load(open(f"{src}/bench.json")), open(f"{dst}/{tag}_bench.json", "w"), indent=1)
for k in ("sr::render_forward_kernel", "sr::render_backward_kernel", "sr::preprocess_backward_kernel", "sr::preprocess_forward_kernel"):
    print(k, out.get(k), sq.get(k))
