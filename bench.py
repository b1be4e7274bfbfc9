#!/usr/bin/env python
"""Benchmark of the hot path: one full surfel-rasterizer forward + backward (K1..K8) per step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (default = BASELINE.json configs[2], the one the metric is quoted on): 3 M synthetic Gaussians,
1920x1080, SH degree 3, all 7 aux-map gradients live, inputs resident in HBM before the timed region.
`--config c2 | c3 | c5` selects a BASELINE configuration as a whole (sizes, live gradients, label); explicit --gaussians /
--width / --height / --no-aux give a "custom" workload that is labelled as such.
N > 1: frames shard one-per-GPU (camera k yawed; --frames-per-rank K: K per GPU, gradients accumulated), then ONE gradient
exchange per step over RCCL -- weak scaling, value = P * frames / s summed over ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector FP32 peak = 64 FLOP/clk/SIMD (plain v_fma_f32, 2 cycles per wave64) -- MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


# BASELINE.json configs: (Gaussians, width, height, aux gradients live, what BASELINE says)
CONFIGS = {
    "c2": (500_000, 1920, 1080, False, "C2: 500k synthetic Gaussians, 1920x1080, SH degree 3, fwd+bwd on 1xMI355X (colour + alpha gradients live)"),
    "c3": (3_000_000, 1920, 1080, True, "C3: 3M Gaussians, 1920x1080, fwd+bwd with depth/normal aux outputs (all 7 aux-map gradients live)"),
    "train_step": (3_000_000, 1920, 1080, True, "train_step: one training view of the reference's late iterations (render + render_semantic + 5 class-filtered rend_dist) as ONE plan, C3 scene"),
    "c5": (6_000_000, 3840, 2160, True, "C5 scene on the GPUs given: 6M Gaussians, 3840x2160, SH degree 3 (BASELINE quotes it on 8 GPUs with a tile sweep: tools/tile_sweep.py)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # SURVEY 8d timing protocol: 10 warm-up + 50 timed
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="a BASELINE configuration as a whole (default: c3)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-aux", action="store_true", help="only colour + alpha gradients live (as in C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the untimed `train_step` section (the reference's 8 rasterizations of a late training iteration vs 2)")
    ap.add_argument("--row-mapped", action="store_true", help="A/B switch: force the row-mapped forward blend (SR_FLAG_ROW_MAPPED_FORWARD; bit-identical results)")
    ap.add_argument("--backward-kernel", choices=["one_wave", "coop", "rows"], default=None, help="A/B switch: force the one-wave-per-tile / the cooperative blend backward (default: by tile count)")
    ap.add_argument("--quadrant-mapped", action="store_true", help="A/B switch: force the quadrant-mapped forward blend (default: picked per frame on the device)")
    ap.add_argument("--exchange", choices=["factored", "allreduce", "compacted"], default="factored",
                    help="N > 1 gradient exchange: factored = all-gather 12-B colour gradients + local SH expansion + all-reduce of "
                         "the other 40 B/Gaussian; allreduce = plain all-reduce of all 232 B/Gaussian; compacted = all-gather of the visibility "
                         "bitmaps (P / 8 B) + all-reduce of the 232 B of the rows reached on at least one rank (parallel.allreduce_visible_rows)")
    ap.add_argument("--rig", choices=["benchmark", "inside"], default="benchmark",
                    help="N > 1 camera rig: benchmark = SURVEY 8d's yawed cameras at the origin (each sees ~86 %% of the Gaussians); inside = cameras "
                         "standing INSIDE the cloud, 80 %% of the Gaussians behind them (synthetic.posed_rig: what a street scene looks like)")
    ap.add_argument("--frames-per-rank", type=int, default=1,
                    help="gradient accumulation: K frames (cameras) per rank and step, ONE gradient exchange per step -- the colour gradients of "
                         "frame j travel while frame j + 1 is computed (factored exchange); value counts all K * N frames")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-oracle work for the cpu_baseline leg")
    a = ap.parse_args()
    custom = any(v is not None for v in (a.gaussians, a.width, a.height)) or (a.no_aux and a.config is None)
    P, W, H, aux, label = CONFIGS[a.config or "c3"]
    a.gaussians = a.gaussians if a.gaussians is not None else P
    a.width = a.width if a.width is not None else W
    a.height = a.height if a.height is not None else H
    a.no_aux = a.no_aux or not aux
    a.tag = "custom" if custom and (a.gaussians, a.width, a.height, not a.no_aux) != (P, W, H, aux) else (a.config or "c3")
    a.label = label if a.tag != "custom" else (f"custom: {a.gaussians} synthetic Gaussians, {a.width}x{a.height}, SH degree {a.sh_degree}, fwd+bwd, "
                                               f"{'colour+alpha' if a.no_aux else 'all 7 aux-map'} gradients live")
    return a


def algorithmic_bytes(P, V, D, npx, deg):
    """SURVEY.md 8(d) 'Algorithmic bytes' per frame, per stage."""
    S = 12 * (deg + 1) ** 2
    return {
        "preprocess": V * (12 + 8 + 16 + 4 + S) + P * 16 + V * 87,
        "binning": D * (12 + 8) + D * 24,
        "blend_fwd": D * 76 + npx * 60,
        "blend_bwd": D * 76 + D * 72 * 2 + npx * (60 + 40),
        "preprocess_bwd": V * (232 + 36 + 72 + 240),
    }


def profiled(kind, args):
    """Latest committed rocprofv3 summary of THIS workload (profiles/rNN_<tag>_<kind>.json, e.g. r02_c3_hbm_traffic.json; the
    round-1 files carry no tag and are C3).  PMC counters cannot be read from inside the timed run, so bench.py quotes the last
    profiled value of the same command; None for a workload that has no committed profile."""
    if args.tag == "custom" or args.sh_degree != 3:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.tag}_{kind}.json")))
    if not files and args.tag == "c3":
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        rel = os.path.relpath(files[-1], ROOT)
        # which build the counters were taken from: PMC passes cannot run inside the timed run, so this is a QUOTE of a committed
        # profile of the same command -- round, HEAD and source digest of that profile are part of the citation.  A profile of OTHER
        # kernel sources is not quoted at all: csrc/, include/ or the build flags changed since (streetunveiler_amd.build.source_digest)
        from streetunveiler_amd.build import source_digest
        now = source_digest()
        src = {"file": rel, "round": d.get("round", os.path.basename(rel).split("_")[0]), "head": d.get("head"), "source_digest": d.get("source_digest"),
               "quoted": "committed rocprofv3 counter pass of the same bench.py command, not measured in this run"}
        if d.get("source_digest") != now:
            src["stale"] = f"the kernels' sources changed since this profile was taken (digest now {now}): counters not quoted -- re-run tools/profile_round.sh"
            return None, src
        return d["kernels"], src
    except Exception:
        return None, None


def measured_ceiling():
    """What simple streaming kernels reach on an MI355X of this pool (tools/ubench/hbm_ceiling.hip -> profiles/rNN_hbm_ceiling.txt: read-only,
    write-only, copy, triad with 16-B accesses, hipMemcpyDtoD; best over grid sizes) -- the practical ceiling next to the 8 TB/s vendor peak."""
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_ceiling.txt")))
    if not files:
        return None
    try:
        txt = open(files[-1]).read()
        rows = re.findall(r"read ([0-9.]+) TB/s\s+write ([0-9.]+)\s+copy ([0-9.]+)\s+triad ([0-9.]+)", txt)
        out = {k: max(float(r[i]) for r in rows) for i, k in enumerate(("read", "write", "copy", "triad"))}
        m = re.search(r"hipMemcpyDtoD[^:]*:\s*([0-9.]+)", txt)
        if m:
            out["hipMemcpyDtoD"] = float(m.group(1))
        out.update(unit="TB/s", source=os.path.relpath(files[-1], ROOT), quoted="committed micro-benchmark run, not measured in this run")
        return out
    except Exception:
        return None


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel`: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, corrected as MI355X_MICROARCH.md
    prescribes (tools/collect_profiles.py)."""
    d, src = profiled("hbm_traffic", args)
    try:
        return d["sr::" + kernel]["traffic_bytes_per_launch"], src
    except Exception:
        return None, src if src and "stale" in src else None


# Per-instruction issue costs FITTED to a micro-benchmark (tools/ubench/valu_issue_ubench.hip, >= 2 waves per SIMD,
# profiles/r02_valu_issue_ubench.txt): plain FP32 ops 1.05-1.3 ns, an SGPR source operand 1.75, DPP 1.8, v_rcp / v_exp 3.4,
# v_permlane32_swap 3.9.  Round 4 measured the clock those kernels actually run at (GRBM_GUI_ACTIVE: 2.1-2.3 GHz for the blend kernels, DVFS):
# 1.05 ns IS the datasheet's 2 cycles at the throttled clock, so `fitted_issue_cost_frac` below is NOT a speed-of-light figure -- it prices
# the instruction mix with costs that already contain the micro-benchmark's own stalls.  `datasheet_issue_frac` is the honest one.
VALU_NS = {"plain": 1.25, "trans": 3.4, "dpp": 1.8, "swap": 3.9}
# cross-lane instructions of the blend backward's per-entry wave reduction (csrc/blend_common.h wave_reduce24), per list entry that
# gets a gradient record: v_permlane32_swap / v_permlane16_swap, and DPP moves / DPP-fused adds
K7_REDUCTION = {"swap": 5, "dpp": 40}


def valu_issue_roof(kernel, ms, args, entries_with_record=0):
    """Where a blend kernel stands against the vector unit, from the committed SQ counter passes of the same command:
      datasheet_issue_frac   VALU instructions x 2 cycles (wave64 on a 32-lane SIMD, MI355X_MICROARCH.md) / 1024 SIMDs / the cycles of
                             the launch at the clock the kernel was MEASURED to run at (GRBM_GUI_ACTIVE / 8 XCDs / its duration);
      valu_busy_frac         SQ_ACTIVE_INST_VALU (quad-cycles with a VALU instruction in flight, summed over the waves; a SIMD takes two
                             per quad-cycle, from two different waves) / 2 / the launch's quad-cycles per SIMD -- the same fraction as the
                             counters see it, transcendentals counted at their double occupancy;
      wave_time              what the resident waves did with their time: issuing an instruction (any kind), stalled at issue (operand
                             dependencies, the VALU port taken by another wave, VALU<->SALU hand-offs), or waiting (s_waitcnt: LDS / memory);
      issue_slots            a wave issues at most one instruction per quad-cycle, whatever its kind: the share of its slots that went to
                             the vector unit -- the scalar, LDS and branch instructions of the walk take the rest;
      waves_per_simd         resident waves averaged over the launch (the tail of a 2.66-round launch included).
    VALU rate = waves_per_simd x issuing x VALU share of the slots, against 2 per quad-cycle.  `fitted_issue_cost_frac` is the round-2/3
    figure (instruction mix priced with micro-benchmark costs), kept for continuity and named for what it is."""
    d, src = profiled("sq_counters", args)
    try:
        k = d["sr::" + kernel]
        n, trans = k["SQ_INSTS_VALU"], k.get("SQ_INSTS_VALU_TRANS_F32", 0)
        swaps = K7_REDUCTION["swap"] * entries_with_record if kernel == "render_backward_kernel" else 0
        dpp = K7_REDUCTION["dpp"] * entries_with_record if kernel == "render_backward_kernel" else 0
        floor_ms = ((n - trans - swaps - dpp) * VALU_NS["plain"] + trans * VALU_NS["trans"] + swaps * VALU_NS["swap"] + dpp * VALU_NS["dpp"]) / 1024 * 1e-6
        out = {"valu_insts_per_launch": int(n), "transcendental": int(trans), "lane_swaps": int(swaps), "dpp": int(dpp),
               "ns_per_inst_per_simd": round(ms * 1e6 * 1024 / n, 3)}
        clock = k.get("clock_ghz")
        if clock:
            cycles = ms * 1e-3 * clock * 1e9                       # per SIMD, this run's duration at the profiled clock
            out["clock_ghz"] = clock
            out["datasheet_issue_frac"] = round(n * 2.0 / 1024 / cycles, 4)
            prof_cycles = k["GRBM_GUI_ACTIVE"] / 8.0                # the profiled launch itself
            if "SQ_ACTIVE_INST_VALU" in k:
                out["valu_busy_frac"] = round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024 / prof_cycles / 2.0, 4)
            if "SQ_WAVE_CYCLES" in k:
                out["waves_per_simd"] = round(k["SQ_WAVE_CYCLES"] * 4.0 / 1024 / prof_cycles, 2)
        if all(c in k for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")):
            w = float(k["SQ_WAVE_CYCLES"])
            out["wave_time"] = {"issuing": round(k["SQ_ACTIVE_INST_ANY"] / w, 3), "stalled_at_issue": round(k["SQ_WAIT_INST_ANY"] / w, 3),
                                "waiting": round(k["SQ_WAIT_ANY"] / w, 3)}
            out["issue_slots"] = {"valu": round(k.get("SQ_ACTIVE_INST_VALU", n) / k["SQ_ACTIVE_INST_ANY"], 3),
                                  "salu": round(k.get("SQ_ACTIVE_INST_SCA", 0) / k["SQ_ACTIVE_INST_ANY"], 3),
                                  "lds": round(k.get("SQ_ACTIVE_INST_LDS", 0) / k["SQ_ACTIVE_INST_ANY"], 3)}
        out.update({"fitted_issue_cost_ms": round(floor_ms, 4), "fitted_issue_cost_frac": round(floor_ms / ms, 4), "source": src,
                    "how": "datasheet_issue_frac = SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / (launch duration x clock_ghz); valu_busy_frac, wave_time, "
                           "issue_slots, waves_per_simd from SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES (quad-cycles) of the committed counter passes; "
                           "fitted_issue_cost_frac = instructions x micro-benchmark cost (plain 1.25 ns, transcendental 3.4, swap 3.9, DPP 1.8: "
                           "profiles/r02_valu_issue_ubench.txt), NOT a speed-of-light figure"})
        return out
    except Exception:
        return {"source": src} if src and "stale" in src else None


def cpu_baseline(args, g, cam, dc, da):
    """The CPU oracle (a port: the reference has no CPU rasterizer) timed on a bounded sub-sample of the
    same scene, forward + backward, OpenMP over all host threads."""
    from oracle import surfel_oracle as so
    native = so.use_native_build()   # -O3 -march=native, compiled on THIS machine; False: no compiler here, the portable -O3 build
    threads = so.num_threads()
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.numpy(), bg=np.zeros(3, np.float32), image_width=args.width,
              image_height=args.height, sh_degree=args.sh_degree,
              tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    best = None
    spent = 0.0
    for n in (25_000, 100_000, 400_000, 1_600_000, args.gaussians):
        n = min(n, args.gaussians)
        sub = {k: v[:n].numpy() for k, v in g.items()}
        t0 = time.perf_counter()
        fwd = so.rasterize_forward(sub["means3D"], sub["opacities"], sub["scales"], sub["rotations"], shs=sub["shs"], **kw)
        so.rasterize_backward(fwd, dc.numpy(), da.numpy())
        dt = time.perf_counter() - t0
        spent += dt
        best = (n, dt, fwd["num_rendered"])
        nxt = {25_000: 100_000, 100_000: 400_000, 400_000: 1_600_000}.get(n, args.gaussians)
        if n == args.gaussians or dt * (min(nxt, args.gaussians) / n) + spent > args.cpu_seconds * 1.5:
            break
    n, dt, D = best
    return {"value": n / dt / 1e6, "unit": "Msplats/s", "cores": threads, "kind": "port",
            "sample": (f"oracle/surfel_oracle.c fwd+bwd, " + ("the whole scene" if n == args.gaussians else f"first {n} Gaussians of the same scene")
                       + f" at {args.width}x{args.height} (D={D}), {dt:.1f} s wall, OpenMP {threads} threads, gcc -O3 "
                       + ("-march=native built on this host" if native else "(portable x86-64 build: no compiler on this host)")),
            "cpu_model": _cpu_model()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def train_step_section(args, params, cam, dev, D, V):
    """Untimed extra (like stage_ms): one late training iteration's rasterizer calls on this scene, the reference's way -- render +
    render_semantic (two passes) + five class-filtered renders = 8 operator calls [REF train.py:84-109] -- and as this build's two
    rasterizations (render_and_semantic + render_class_distortions) and as ONE plan (render_train_view: one K1, one binning, one K8); ms per
    fwd+bwd and how far the resulting maps are apart.  Then the one-plan pattern again with the library's stage timers on: per-stage ms of
    one iteration and a `roofline` block for its dominant kernel."""
    from streetunveiler_amd import _lib
    from streetunveiler_amd.gaussian_renderer import SurfelModel
    from streetunveiler_amd.train_pattern import compare_and_time, make_weights, one_plan_pattern
    sem = torch.randint(0, 6, (args.gaussians,), generator=torch.Generator().manual_seed(0)).to(dev)
    sem[sem == 4] = 2   # the reference prunes the sky Gaussians before training
    for t in params.values():
        t.grad = None
    pc = SurfelModel(params["means3D"], params["scales"], params["rotations"], params["opacities"], params["shs"], sem, args.sh_degree, 3)
    try:
        res = compare_and_time(cam.to(dev), pc, torch.zeros(3, device=dev), list(params.values()))
        lib = _lib.load()
        weights = make_weights(args.height, args.width, dev); bg = torch.zeros(3, device=dev); camd = cam.to(dev)
        iters = 3
        torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
        for _ in range(iters):
            for t in params.values():
                t.grad = None
            one_plan_pattern(camd, pc, bg, weights)["loss"].backward()
        torch.cuda.synchronize()
        st = _lib.stage_stats()
        lib.sr_set_stage_timing(0)
        res["one_plan_stage_ms_per_iteration"] = {k: round(ms / iters, 4) for k, (ms, n) in st.items() if n}
        res["roofline"] = train_step_roofline(args, res["one_plan_stage_ms_per_iteration"], D, V, n_classes=5)
    finally:
        for t in params.values():
            t.grad = None
        torch.cuda.empty_cache()
    return res


def train_step_roofline(args, stage_ms, D, V, n_classes):
    """The `roofline` block of the training-step pattern: its dominant kernel against HBM, as for the headline kernel -- algorithmic
    bytes (the per-class pass reads 56 B per list entry -- index + the 52 B of the record it stages -- and its backward leaves
    one 15-float gradient record per entry, counted as K7's are: read-modify-write = 120 B; 20 B per pixel and class of state each way)
    over the launch time measured in this run, the committed counter traffic of the same pattern (profiles/rNN_train_step_*), and the
    vector-issue figures that actually bound it."""
    npx = args.width * args.height
    ab = {"class_fwd": D * 56 + n_classes * npx * 20, "class_bwd": D * 56 + D * 120 + n_classes * npx * 24,
          "blend_fwd": D * (76 + 24) + npx * (60 + 24), "blend_bwd": D * (76 + 24) + D * (72 + 24) * 2 + npx * (60 + 40 + 24)}   # (9-channel pass: six more colour floats per entry, pixel and record)
    kernels = {"class_fwd": "class_forward_kernel", "class_bwd": "class_backward_kernel", "blend_fwd": "render_forward_kernel", "blend_bwd": "render_backward_kernel"}
    cand = {k: stage_ms.get(k) for k in ab if stage_ms.get(k)}
    if not cand:
        return None
    dom = max(cand, key=cand.get)
    prof = argparse.Namespace(tag="train_step", sh_degree=3) if (args.tag == "c3" and args.sh_degree == 3) else argparse.Namespace(tag="custom", sh_degree=args.sh_degree)
    out = {"bound": "hbm", "kernel": kernels[dom], "avg_launch_ms": cand[dom], "algorithmic_bytes_per_launch": ab[dom], "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    out["achieved"] = round(ab[dom] / (cand[dom] * 1e-3) / 1e9, 2); out["frac"] = round(out["achieved"] / HBM_PEAK_GBS, 5)
    out["traffic"], out["traffic_source"] = pmc_traffic(kernels[dom], prof)
    out["per_kernel"] = {}
    for k, ms in cand.items():
        t, _ = pmc_traffic(kernels[k], prof)
        out["per_kernel"][kernels[k]] = {"ms": ms, "algorithmic_bytes": ab[k], "hbm_frac": round(ab[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": t,
                                         "traffic_over_algorithmic": None if not t else round(t / ab[k], 3), "issue_roof": valu_issue_roof(kernels[k], ms, prof)}
    out["note"] = ("the per-class kernels are vector-issue-bound like K6 / K7 (per_kernel.*.issue_roof.datasheet_issue_frac), the HBM fraction is reported for "
                   "comparability with the headline block; duplicates D and visible V as in config")
    return out


XGMI_LINK_GBS = 76.8    # one xGMI link, ONE direction: 7 links per GPU at ~153.6 GB/s bidirectional each (the figure SURVEY.md 8e quotes per link)


def camera_inside_scene_section(P, W, H, deg, dc, da, dev, behind=0.8, steps=10):
    """NOT the headline metric -- reported beside it: the same P Gaussians with the camera standing INSIDE the cloud (a posed camera with
    `behind` of the Gaussians behind it: what a street scene looks like to the operator; the benchmark frame sees 86 % of its Gaussians).
    One fwd+bwd step as in the timed region, after it, on its own scene (streetunveiler_amd.synthetic.posed_scene; tools/time_surrounded.py)."""
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from streetunveiler_amd.synthetic import posed_scene
    cam, g = posed_scene(P, W, H, seed=7, spread=25.0, behind_fraction=behind)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                      cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, False)
    t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)

    def step():
        for v in list(t.values()) + [m2]:
            v.grad = None
        c, r, am = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, am], [dc, da])
        return r
    for _ in range(3):
        r = step()
    V = int((r > 0).sum())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"what": f"{P} Gaussians, {W}x{H}, posed camera with {behind:.0%} of the Gaussians behind it (untimed extra, not the metric)",
            "visible_frac": round(V / P, 3), "ms_per_step": round(ms, 4), "msplats_per_s": round(P / ms / 1e3, 1), "steps": steps}


def rccl_topology(log_path, backend, world):
    """What the communicator says it built: channel count and ring / tree lines of RCCL's own INIT / GRAPH log (NCCL_DEBUG_FILE, this rank)."""
    import re
    out = {"backend": backend, "channels": None, "log": log_path}
    if backend != "nccl" or not log_path or not os.path.exists(log_path):
        return out
    try:
        txt = open(log_path, errors="replace").read()
    except OSError:
        return out
    ch = [int(b) for _, b in re.findall(r"Channel (\d+)/(\d+)", txt)]
    if ch:
        out["channels"] = max(ch)
    m = re.findall(r"(\d+) coll channels", txt)
    if m:
        out["coll_channels"] = int(m[-1]); out["channels"] = out["channels"] or int(m[-1])
    m = re.search(r"nranks (\d+)", txt)
    if m:
        out["nranks_in_log"] = int(m.group(1))
    out["rings_logged"] = len(re.findall(r"Ring \d+", txt)); out["trees_logged"] = len(re.findall(r"Trees? \[", txt))
    ver = re.search(r"(RCCL version [^\n]+|NCCL version [^\n]+)", txt)
    if ver:
        out["version"] = ver.group(1).strip()[:80]
    return out


def isolated_collectives(P, world, dev, exchange):
    """The step's collectives alone, nothing else on the GPU (after the timed region): the all-gather of the 12-B colour gradients and the
    all-reduce of the other 40 B/Gaussian (factored exchange), or the all-reduce of all 232 B/Gaussian -- ms, bytes and bandwidth each.
    busbw as nccl-tests define it: all-reduce 2 (N - 1) / N x bytes / t, all-gather (N - 1) / N x total bytes / t."""
    out = {}

    def timed(fn, iters=5):
        fn(); torch.cuda.synchronize(); dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters
    n = max(world, 1)
    if exchange == "factored":
        gc = torch.zeros(P * 3, device=dev); gathered = torch.empty(P * 3 * world, device=dev)
        ms = timed(lambda: dist.all_gather_into_tensor(gathered, gc))
        total = gathered.numel() * 4
        out["all_gather_colour_gradients"] = {"bytes_per_rank": P * 12, "bytes_gathered": total, "ms": round(ms, 4),
                                              "algbw_GBs": round(total / ms / 1e6, 2), "busbw_GBs": round(total * (n - 1) / n / ms / 1e6, 2)}
        rest = torch.zeros(P * 10, device=dev)
        del gc, gathered
    else:
        rest = torch.zeros(P * 58, device=dev)
    ms = timed(lambda: dist.all_reduce(rest))
    nbytes = rest.numel() * 4
    out["all_reduce_rest"] = {"bytes": nbytes, "ms": round(ms, 4), "algbw_GBs": round(nbytes / ms / 1e6, 2),
                              "busbw_GBs": round(nbytes * 2 * (n - 1) / n / ms / 1e6, 2)}
    return out


def predicted_xgmi(P, world, exchange):
    """Wire time of the step's collectives from the link figures alone (no latency, perfect pipelining): a ring moves its payload over ONE
    link per hop (busbw = one link), a direct exchange on the fully connected 8-GPU xGMI mesh uses the N - 1 links of a GPU at once."""
    n = max(world, 1)
    if n == 1:
        return {"all_gather_ms": 0.0, "all_reduce_ms": 0.0, "note": "one rank: nothing crosses a link"}
    ag_bytes = P * 12 * n if exchange == "factored" else 0
    ar_bytes = P * (40 if exchange == "factored" else 232)
    ring = lambda moved: moved / (XGMI_LINK_GBS * 1e9) * 1e3
    return {"link_GBs": XGMI_LINK_GBS,
            "ring": {"all_gather_ms": round(ring(ag_bytes * (n - 1) / n), 3), "all_reduce_ms": round(ring(ar_bytes * 2 * (n - 1) / n), 3)},
            "direct_all_links": {"all_gather_ms": round(ring(ag_bytes * (n - 1) / n) / (n - 1), 3), "all_reduce_ms": round(ring(ar_bytes * 2 * (n - 1) / n) / (n - 1), 3)},
            "note": "ring = per-link bound (what a single RCCL ring reaches); direct = every peer link busy at once (the mesh's ceiling); RCCL with several "
                    "channels lands between the two"}


def exchange_report(detail, P, world, K, ms_per_step, exchange):
    """The first N > 1 run has to explain itself: per-rank step times, how much of the exchange the compute stream actually waited for,
    what the same collectives cost alone, and what the links would allow."""
    if detail is None:
        return None
    iso = detail.get("isolated_collectives", {})
    # per step: one all-gather per frame of the rank (K of them with gradient accumulation), one all-reduce
    alone = sum(v["ms"] * (K if name == "all_gather_colour_gradients" else 1) for name, v in iso.items())
    exposed = max(detail["exposed_ms_per_step_per_rank"]) if detail["exposed_ms_per_step_per_rank"] else 0.0
    out = dict(detail)
    out["collectives_alone_ms_per_step"] = round(alone, 4)
    out["exposed_ms_per_step_max_over_ranks"] = round(exposed, 4)
    out["hidden_fraction_of_the_collectives"] = None if alone <= 0 else round(max(0.0, 1.0 - exposed / alone), 4)
    out["exposed_fraction_of_the_step"] = round(exposed / ms_per_step, 4) if ms_per_step else None
    out["predicted_xgmi"] = predicted_xgmi(P, world, exchange)
    return out


def _flush_c_stdio():
    import ctypes
    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _discard_further_stdout():
    try:
        sys.stdout.flush()
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)
        os.close(null)
    except OSError:
        pass


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N workers ourselves -- one process per GPU under
    torch.distributed.run on the loopback address, same flags -- and hand their output through.  (The driver wraps its N > 1 runs in
    torch.distributed.run itself; this makes the bare command line of the 1-GPU run work for any N.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a torchrun environment: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def reference_operating_point_section(deg, dev, P=1_500_000, W=480, H=320, steps=20):
    """NOT the headline metric -- reported beside it: the reference's DOCUMENTED operating point.  Every example run of the reference uses
    `-r 4` [REF /root/reference/README.md:195-207]: Waymo's 1920x1280 frames rendered at 480x320, with a Waymo-segment-sized model.  At
    that size a step is a few hundred microseconds of kernels, and what the host does between them shows: the forward's read-back of the
    duplicate count (the reference waits there too) keeps the host from queueing ahead.  Reported: fwd+bwd per step in the default mode (which
    below 2 600 tiles runs the cooperative blend backward), with the one-wave backward forced, with 8x8 tiles (`tile="auto"`), with a
    binning capacity (SR_FLAG_BINNING_CAPACITY: no host wait anywhere), per-stage ms, and -- on a 10 k-Gaussian frame -- the forward's
    floor in the three ways of calling it (default / capacity / capacity inside a HIP graph)."""
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    from streetunveiler_amd import _lib
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
    lib = _lib.load()
    out = {"what": f"{P} Gaussians at {W}x{H} (the reference's `-r 4` frames), fwd+bwd, all aux gradients live (untimed extra, not the metric)"}
    e = torch.empty(0, device=dev)

    def scene(P, W, H):
        cam = synthetic_camera(W, H); g = synthetic_gaussians(P, W, H, seed=0)
        s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                          cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, False)
        t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
        dc, da = (x.to(dev) for x in synthetic_upstream_grads(W, H, seed=1))
        return s, t, torch.zeros(P, 3, device=dev, requires_grad=True), dc, da

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    s, t, m2, dc, da = scene(P, W, H)

    def step(r):
        for v in list(t.values()) + [m2]:
            v.grad = None
        c, radii, am = r(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, am], [dc, da])
    r0 = GaussianRasterizer(s)
    with torch.no_grad():
        D = int(_C.rasterize_gaussians(s.bg, t["means3D"].detach(), e, t["opacities"].detach(), t["scales"].detach(), t["rotations"].detach(), 1.0, e, s.viewmatrix,
                                       s.projmatrix, s.tanfovx, s.tanfovy, H, W, t["shs"].detach(), deg, s.campos, False, False)[0])
    rc = GaussianRasterizer(s, binning_capacity=int(1.25 * D) + 1024)
    ra = GaussianRasterizer(s, tile="auto")   # (optional: 8x8 tiles here -- moves the algorithm's tile-granular 3-sigma truncation)
    out["duplicates_D"] = D
    out["ms_per_step_default"] = round(timed(lambda: step(r0), steps), 4)          # below 2 600 tiles: the four-wave cooperative blend backward
    out["msplats_per_s_default"] = round(P / out["ms_per_step_default"] / 1e3, 1)
    out["ms_per_step_one_wave_backward"] = round(timed(lambda: step(GaussianRasterizer(s, backward_kernel="one_wave")), steps), 4)
    out["ms_per_step_binning_capacity"] = round(timed(lambda: step(rc), steps), 4)
    out["overflowed"] = int(rc.last_status.tolist()[2])
    out["tile_auto"] = list(ra.tile or (16, 16))
    out["ms_per_step_tile_auto"] = round(timed(lambda: step(ra), steps), 4)
    out["msplats_per_s_tile_auto"] = round(P / out["ms_per_step_tile_auto"] / 1e3, 1)
    rc = r0   # per-stage ms of the default
    torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
    for _ in range(3):
        step(rc)
    torch.cuda.synchronize()
    st = _lib.stage_stats(); lib.sr_set_stage_timing(0)
    out["stage_ms"] = {k: round(ms / max(n, 1), 4) for k, (ms, n) in st.items() if n}
    out["stage_ms_sum"] = round(sum(out["stage_ms"].values()), 4)
    del t, m2, dc, da, r0, rc, ra
    # ---- the forward's floor on a small frame: 10 k Gaussians, 256x256, under no_grad (an inference loop)
    s, t, m2, dc, da = scene(10_000, 256, 256)
    args_ = lambda: (s.bg, t["means3D"].detach(), e, t["opacities"].detach(), t["scales"].detach(), t["rotations"].detach(), 1.0, e, s.viewmatrix, s.projmatrix,
                     s.tanfovx, s.tanfovy, 256, 256, t["shs"].detach(), deg, s.campos, False, False)
    with torch.no_grad():
        Ds = int(_C.rasterize_gaussians(*args_())[0])
        cap = int(1.5 * Ds) + 1024
        floor = {"default_us": round(1e3 * timed(lambda: _C.rasterize_gaussians(*args_(), forward_only=True), 200), 1),
                 "binning_capacity_us": round(1e3 * timed(lambda: _C.rasterize_gaussians(*args_(), forward_only=True, binning_capacity=cap), 200), 1)}
        try:
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _C.rasterize_gaussians(*args_(), forward_only=True, binning_capacity=cap)
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                keep = _C.rasterize_gaussians(*args_(), forward_only=True, binning_capacity=cap)
            floor["binning_capacity_in_a_hip_graph_us"] = round(1e3 * timed(graph.replay, 200), 1)
            del keep, graph
        except Exception as ex:   # noqa: BLE001 -- reported, not fatal: the section is an extra
            floor["binning_capacity_in_a_hip_graph_us"] = None; floor["graph_error"] = f"{type(ex).__name__}: {ex}"[:200]
    out["forward_floor_10k_gaussians_256x256"] = floor
    return out


def train_step_main(args):
    """`--config train_step`: the work the reference actually does per late training iteration [REF /root/reference/train.py:84-109] as the
    TIMED region -- `render_train_view` (9-channel render + five per-class distortion maps on ONE preprocess / binning / per-Gaussian
    backward) forward + backward incl. the allmap post-processing and the loss kernels, on the C3 scene.  One JSON line with the contract's
    fields; `roofline` = its dominant kernel measured with HIP event pairs on the launch stream inside extra steps; `reference_8_calls_ms` =
    the same maps the reference's way (8 operator calls through this build) beside it."""
    assert args.gpus == 1, "--config train_step is a single-GPU mode (the frame-parallel exchange is measured by the default config)"
    from streetunveiler_amd import _lib
    from streetunveiler_amd.gaussian_renderer import SurfelModel
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
    from streetunveiler_amd.train_pattern import make_weights, one_plan_pattern, reference_pattern
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    lib = _lib.load()
    P, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    g = synthetic_gaussians(P, W, H, seed=0)
    params = {k: v.to(dev).requires_grad_() for k, v in g.items()}
    sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(0)).to(dev)
    sem[sem == 4] = 2   # the reference prunes the sky Gaussians before training
    pc = SurfelModel(params["means3D"], params["scales"], params["rotations"], params["opacities"], params["shs"], sem, deg, 3)
    cam = synthetic_camera(W, H).to(dev)
    weights = make_weights(H, W, dev); bg = torch.zeros(3, device=dev)
    leaves = list(params.values())

    def step(fn=one_plan_pattern):
        for t in leaves:
            t.grad = None
        m = fn(cam, pc, bg, weights)
        m["loss"].backward()
        return m
    for _ in range(max(args.warmup, 1)):
        m = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    # per-stage ms on extra steps (HIP event pairs recorded by the library on the launch stream)
    iters = 5
    lib.sr_set_stage_timing(1)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    st = _lib.stage_stats(); lib.sr_set_stage_timing(0)
    stage_ms = {k: round(v / iters, 4) for k, (v, n) in st.items() if n}
    # scene statistics
    with torch.no_grad():
        from diff_surfel_rasterization import _C
        e = torch.empty(0, device=dev)
        D, _, _, radii0, *_ = _C.rasterize_gaussians(bg, params["means3D"].detach(), e, params["opacities"].detach(), params["scales"].detach(), params["rotations"].detach(),
                                                     1.0, e, cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W,
                                                     params["shs"].detach(), deg, cam.camera_center, False, False)
        V = int((radii0 > 0).sum())
    ref_ms = None
    if not args.no_train_step:   # the reference's eight calls beside it (untimed extra)
        for _ in range(2):
            step(reference_pattern)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            step(reference_pattern)
        torch.cuda.synchronize(); ref_ms = round((time.perf_counter() - t0) * 1e3 / 3, 3)
    prof_args = argparse.Namespace(**{**vars(args), "tag": "c3"})
    roof = train_step_roofline(prof_args, stage_ms, D, V, n_classes=5)
    out = {"metric": "Msplats/s, one training view fwd+bwd (9-channel render + 5 per-class distortion maps on one plan) @1920x1080, 3M Gaussians",
           "value": round(P / ms / 1e3, 3), "unit": "Msplats/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "train_step: the rasterizer work of one late training iteration of the reference [train.py:84-109] on the C3 scene -- render + "
                                  "render_semantic + five class-filtered rend_dist maps -- as render_train_view (one K1, one binning, one K8); incl. allmap post-processing and loss kernels",
                      "baseline_config": "train_step", "gaussians": P, "width": W, "height": H, "sh_degree": deg, "visible": V, "duplicates_D": int(D), "parallelism": "single GPU"},
           "roofline": roof, "stage_ms": stage_ms, "stage_ms_sum": round(sum(stage_ms.values()), 4),
           "reference_8_calls_ms": ref_ms, "speedup_over_the_reference_call_pattern": None if not ref_ms else round(ref_ms / ms, 2)}
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.config == "train_step":
        return train_step_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    from streetunveiler_amd import parallel as par
    from streetunveiler_amd.parallel import allreduce_gradients, factored_sh_exchange, init_distributed
    rccl_log = None
    if args.gpus > 1 or os.environ.get("SURFEL_EXCHANGE_SINGLE_RANK") == "1":
        # the communicator's own account of what it built (rings / channels), one file per rank: read back after the warm-up collectives
        rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"sr_rccl_{os.getpid()}.log")
        os.environ.setdefault("NCCL_DEBUG", "INFO"); os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH"); os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        rccl_log = os.environ["NCCL_DEBUG_FILE"].replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)
    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: start {args.gpus} ranks "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or drop the environment")
    # SURFEL_EXCHANGE_SINGLE_RANK=1 (with torchrun --nproc-per-node 1): take the frame-parallel branches on a one-rank RCCL group -- the
    # same calls an N-GPU run makes, exercised on a single GPU (functional check; the JSON line then reports n_gpus 1 with an exchange)
    multi = world > 1 or os.environ.get("SURFEL_EXCHANGE_SINGLE_RANK") == "1"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())  # (the modulo only matters for the gloo smoke test)
    torch.cuda.set_device(dev)

    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    from streetunveiler_amd import _lib
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
    lib = _lib.load()

    P, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    g_cpu = synthetic_gaussians(P, W, H, seed=0)
    K = max(1, args.frames_per_rank)
    n_cams = world * K
    # frame j of rank r = camera r * K + j of the yawed camera batch (one camera, the unrotated one, for the plain single-GPU run)
    rig_cams = None
    if args.rig == "inside":   # cameras in general position inside the cloud; every rank builds the same rig and Gaussians
        from streetunveiler_amd.synthetic import posed_rig
        rig_cams, g_cpu = posed_rig(P, W, H, max(n_cams, 1), seed=7, spread=25.0, behind_fraction=0.8)
    cam_of = (lambda i: rig_cams[i]) if rig_cams is not None else (lambda i: synthetic_camera(W, H) if n_cams == 1 else synthetic_camera(W, H, index=i, n_cams=n_cams))
    cams = [cam_of(rank * K + j) for j in range(K)]
    cam = cams[0]
    dc_cpu, da_cpu = synthetic_upstream_grads(W, H, seed=1, aux=not args.no_aux)
    params = {k: v.to(dev).requires_grad_() for k, v in g_cpu.items()}
    dc, da = dc_cpu.to(dev), da_cpu.to(dev)
    make_settings = lambda c: GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), torch.zeros(3, device=dev),
                                                            1.0, c.world_view_transform.to(dev), c.full_proj_transform.to(dev), deg,
                                                            c.camera_center.to(dev), False, False)
    settings = make_settings(cam)
    rasterizers = [GaussianRasterizer(settings if j == 0 else make_settings(cams[j]), row_mapped=True if args.row_mapped else (False if args.quadrant_mapped else None), backward_kernel=args.backward_kernel) for j in range(K)]
    # the camera list is replicated: every rank knows every rank's camera positions ([world, 3], or [world, K, 3] with accumulation)
    all_campos = None
    if multi:
        all_campos = torch.stack([torch.stack([cam_of(r * K + j).camera_center for j in range(K)])
                                  for r in range(world)]).to(dev)
        if K == 1:
            all_campos = all_campos[:, 0]
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    leaves = [params["means3D"], params["shs"], params["opacities"], params["scales"], params["rotations"], means2D]

    exchange_log = {"ms": 0.0, "bytes": 0, "calls": 0, "early_starts": 0}

    def frame(j):   # one frame: operator forward + backward; parameter gradients accumulate in the leaves
        color, radii, allmap = rasterizers[j](means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                              opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward([color, allmap], [dc, da])
        return radii

    def step():     # K frames per rank, then (N > 1) ONE gradient exchange
        for t in leaves:
            t.grad = None
        if multi and args.exchange == "factored":
            # SH gradient: all-gather of the 12-B colour gradients (per frame, started between K7 and K8, travelling under the next
            # frame's kernels) + local expansion of all K * N views; the other 40 B/Gaussian: one all-reduce -- inside the last
            # backward for K = 1 (every gradient leaves backward summed over the ranks), after the K-th frame otherwise
            with factored_sh_exchange(all_campos=all_campos, reduce_all=K == 1, frames_per_rank=K) as ex:
                for j in range(K):
                    radii = frame(j)
                if K > 1:
                    ex.finish([t.grad for t in (leaves[0], leaves[2], leaves[3], leaves[4])])
            exchange_log["ms"] += ex.exchange_ms; exchange_log["bytes"] += ex.bytes_sent; exchange_log["calls"] += ex.calls
            exchange_log["early_starts"] += ex.early_starts
        else:
            seen = None
            for j in range(K):
                radii = frame(j)
                seen = (radii > 0) if seen is None else (seen | (radii > 0))   # (the rows this rank's frames reached)
            if multi and args.exchange == "compacted":
                t0 = time.perf_counter()
                info = par.allreduce_visible_rows([t.grad for t in leaves[:5]], seen)   # the rows reached on at least one rank only
                exchange_log["ms"] += (time.perf_counter() - t0) * 1e3; exchange_log["calls"] += 1
                exchange_log["bytes"] += info["bytes_payload"]; exchange_log["rows_union"] = info["rows_union"]
            elif multi:
                t0 = time.perf_counter()
                allreduce_gradients([t.grad for t in leaves[:5]])   # 232 B/Gaussian, one collective over the flat buffer
                exchange_log["ms"] += (time.perf_counter() - t0) * 1e3; exchange_log["calls"] += 1
                exchange_log["bytes"] += sum(t.grad.numel() * 4 for t in leaves[:5])
        return radii

    # scene statistics (outside the timed region)
    with torch.no_grad():
        e = torch.empty(0, device=dev)
        D, _, _, radii0, *_ = _C.rasterize_gaussians(settings.bg, params["means3D"].detach(), e, params["opacities"].detach(),
                                                     params["scales"].detach(), params["rotations"].detach(), 1.0, e,
                                                     settings.viewmatrix, settings.projmatrix, settings.tanfovx, settings.tanfovy,
                                                     H, W, params["shs"].detach(), deg, settings.campos, False, False)
        V = int((radii0 > 0).sum().item())
    del radii0

    # what the blend kernels execute on this scene (device counters of one forward, outside the timed region)
    counters = torch.zeros(16, dtype=torch.int64, device=dev)
    with torch.no_grad():
        GaussianRasterizer(settings, blend_counters=counters)(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                                              opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    torch.cuda.synchronize()
    st = counters.tolist()
    blend_counts = {"staged_entries_D_eff": int(st[0]), "entries_after_quadrant_cull": int(st[1]), "quadrant_tests": int(st[2]),
                    "quadrant_tests_with_a_hit": int(st[3]), "contributing_pairs": int(st[4]), "entries_with_a_hit": int(st[7]),
                    "cell_pairs_4x4": int(st[8]), "row_mapping_steps": int(st[9]),
                    "cell_pairs_4x4_octagon_culling": int(st[10]), "row_mapping_steps_octagon_culling": int(st[11]),
                    "cell_pairs_4x4_box_culling": int(st[12]), "row_mapping_steps_box_culling": int(st[13]), "box_culling_dropped_hits": int(st[14])}

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_check = None

    worlds_seen = None
    rccl_info = None
    if multi:   # bring the communicator up before anything is timed, whatever --warmup says
        w = torch.zeros(1024, device=dev); wg = torch.empty(1024 * world, device=dev)
        dist.all_reduce(w); dist.all_gather_into_tensor(wg, w)
        sync()
        # what every rank believes the job looks like: its communicator's size (all-gathered over that communicator)
        mine = torch.tensor([float(dist.get_world_size())], device=dev); seen = torch.empty(world, device=dev)
        dist.all_gather_into_tensor(seen, mine)
        worlds_seen = [int(x) for x in seen.tolist()]
        if any(w != args.gpus for w in worlds_seen) and os.environ.get("SURFEL_EXCHANGE_SINGLE_RANK") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the ranks report communicator sizes {worlds_seen}: the job is not ONE {args.gpus}-rank group")
        rccl_info = rccl_topology(rccl_log, dist.get_backend(), world)
        # (decided by ALL ranks together: a rank that left on its own would leave the others waiting in the next collective)
        mine = torch.tensor([float(rccl_info["channels"] if rccl_info.get("channels") is not None else -1)], device=dev); seen = torch.empty(world, device=dev)
        dist.all_gather_into_tensor(seen, mine)
        rccl_info["channels_seen_by_each_rank"] = [int(x) for x in seen.tolist()]
        if world > 1 and any(0 <= c <= 1 for c in rccl_info["channels_seen_by_each_rank"]):
            # a single-channel ring puts the whole exchange on one xGMI link: the number is still a measurement of THIS job, so the run goes on
            # and the JSON line says so (it used to abort here -- on the first real multi-GPU run a mis-read log line would have cost the
            # whole scaling curve); SURFEL_REQUIRE_MULTI_CHANNEL=1 restores the abort
            msg = (f"RCCL's log reads as a single-channel ring for {world} ranks: one xGMI link would carry the whole exchange -- check "
                   f"NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS / the topology file")
            if os.environ.get("SURFEL_REQUIRE_MULTI_CHANNEL") == "1":
                raise SystemExit(f"bench.py: {msg} ({rccl_info})")
            rccl_info["warning"] = msg
            if rank == 0:
                print(f"bench.py: warning: {msg}", file=sys.stderr, flush=True)
        par.STALLS.enabled = True   # compute-stream stalls at the collectives' wait points = the exposed part of the exchange
        if args.exchange == "factored":
            # one untimed trial step of the factored exchange.  Only a failure of the COLLECTIVE LAYER (a backend that lacks
            # all_gather_into_tensor, a communicator error: torch.distributed raises DistBackendError / NotImplementedError, on every
            # rank alike) falls back to the plain all-reduce, and the JSON line carries the message; anything else -- a wrong result
            # included -- is a bug and ends the run.
            fallback_errors = (NotImplementedError,) + ((dist.DistBackendError,) if hasattr(dist, "DistBackendError") else ())
            try:
                step()
                sync()
                # self-check of the collective path on its first real outing (RCCL over xGMI has only ever been exercised by the
                # driver's multi-GPU run): the factored exchange must give the gradients of the plain all-reduce
                fact = [t.grad.clone() for t in leaves[:5]]
                args.exchange = "allreduce"; step(); sync(); args.exchange = "factored"
                worst = max(float((a - t.grad).abs().max() / (t.grad.abs().max() + 1e-30)) for a, t in zip(fact, leaves[:5]))
                exchange_check = {"factored_vs_allreduce_max_rel_err": worst, "ok": bool(worst < 1e-4)}
                if not exchange_check["ok"]:
                    raise RuntimeError(f"factored exchange disagrees with the plain all-reduce ({worst:.2e})")
            except fallback_errors as e:
                print(f"[rank {rank}] factored exchange unavailable ({type(e).__name__}: {e}); using all-reduce", file=sys.stderr, flush=True)
                args.exchange = "allreduce"
                exchange_check = {"ok": False, "fell_back_to": "allreduce", "error": f"{type(e).__name__}: {e}"[:300]}

        if args.exchange == "compacted":   # the rows-of-the-union exchange must give the dense all-reduce's sums (ring order aside)
            step(); sync()
            comp = [t.grad.clone() for t in leaves[:5]]
            args.exchange = "allreduce"; step(); sync(); args.exchange = "compacted"
            worst = max(float((a - t.grad).abs().max() / (t.grad.abs().max() + 1e-30)) for a, t in zip(comp, leaves[:5]))
            exchange_check = {"compacted_vs_allreduce_max_rel_err": worst, "ok": bool(worst < 1e-5)}
            if not exchange_check["ok"]:
                raise RuntimeError(f"visibility-compacted exchange disagrees with the plain all-reduce ({worst:.2e})")

    for _ in range(args.warmup):
        step()
    sync()
    par.STALLS.drain_ms()
    exchange_log.update(ms=0.0, bytes=0, calls=0, early_starts=0)
    # Timed region: HIP events (recorded by the library on the launch stream) around the DOMINANT kernel only -- the blend backward,
    # the kernel of `roofline` -- because every event record costs ~5 us of stream time (tools/step_timeline.py: bracketing both blend
    # kernels cost 30 us per step, all nine stages ~1.4 %).  Every other stage is timed right after the timed region, on extra steps.
    blend_mask = 1 << list(_lib.SR_STAGE_NAMES).index("blend_bwd")
    lib.sr_set_stage_timing(2 * blend_mask)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        step()
        marks[k + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    timed_exchange = dict(exchange_log)
    exposed_ms = par.STALLS.drain_ms() / args.steps if multi else 0.0   # (after sync(): every event pair has completed)
    par.STALLS.enabled = False
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 4)
    stats = _lib.stage_stats()
    lib.sr_set_stage_timing(1)
    for _ in range(min(args.steps, 5)):
        step()
    sync()
    stats_all = _lib.stage_stats()
    lib.sr_set_stage_timing(0)
    stats = {k: (stats[k] if stats[k][1] else stats_all[k]) for k in stats_all}
    exchange_detail = None
    if multi:
        mine = torch.tensor([elapsed / args.steps * 1e3, exposed_ms], device=dev, dtype=torch.float64)
        allr = torch.empty(2 * world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 2).tolist()
        exchange_detail = {"per_rank_ms_per_step": [round(a, 4) for a, _ in allr],
                           "exposed_ms_per_step_per_rank": [round(b, 4) for _, b in allr],
                           "exposed_definition": "time rank r's compute stream stood still at the collectives' wait points (event pairs around work.wait()), per step"}
        exchange_detail["isolated_collectives"] = isolated_collectives(P, world, dev, args.exchange)
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P * world * K * args.steps / elapsed / 1e6
        npx = W * H
        ab = algorithmic_bytes(P, V, D, npx, deg)
        stage_ms = {k: (ms / n if n else None) for k, (ms, n) in stats.items()}
        groups = {"preprocess": ["preprocess"], "binning": ["depth_sort", "scan", "expand_x", "expand_y", "ranges"],
                  "blend_fwd": ["blend_fwd"], "blend_bwd": ["blend_bwd"], "preprocess_bwd": ["preprocess_bwd"]}
        group_ms = {k: sum(stage_ms[s] or 0.0 for s in v) for k, v in groups.items()}
        dominant = max(("blend_fwd", "blend_bwd"), key=lambda k: group_ms[k])  # the north-star kernels
        ach = ab[dominant] / (group_ms[dominant] * 1e-3) / 1e9 if group_ms[dominant] else None
        blend_ms = group_ms["blend_fwd"] + group_ms["blend_bwd"]
        blend_gbs = (ab["blend_fwd"] + ab["blend_bwd"]) / (blend_ms * 1e-3) / 1e9 if blend_ms else None
        kernels_ms = sum(group_ms.values())
        dom_kernel = "render_backward_kernel" if dominant == "blend_bwd" else "render_forward_kernel"
        traffic, traffic_src = pmc_traffic(dom_kernel, args)
        # the streaming kernels against what streaming kernels reach on this pool's boxes (read 6.5, copy 5.0-5.9 TB/s measured; 8 is the vendor peak)
        ceiling = measured_ceiling()
        hbm_bound = {}
        for stage, kern in (("preprocess", "preprocess_forward_kernel"), ("preprocess_bwd", "preprocess_backward_kernel")):
            t, _ = pmc_traffic(kern, args)
            if t and group_ms[stage]:
                gbs = t / (group_ms[stage] * 1e-3) / 1e9
                hbm_bound[kern] = {"traffic_bytes_per_launch": int(t), "ms": round(group_ms[stage], 4), "achieved_GBs": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                                   "frac_of_measured_copy_ceiling": None if not ceiling else round(gbs / 1e3 / ceiling["copy"], 4)}
        # FP32-VALU view of the same two kernels (SURVEY 8d "algorithmic flops": 60 fwd + 200 bwd per tested pair).  K7 walks
        # exactly the (entry, quadrant) pairs K6 recorded, so one set of counters serves both.
        lane_tests = blend_counts["quadrant_tests"] * 64
        useful = blend_counts["contributing_pairs"] * 260 / (blend_ms * 1e-3) / 1e12 if blend_ms else None
        issued = lane_tests * 260 / (blend_ms * 1e-3) / 1e12 if blend_ms else None
        # which mapping the forward blend ran (the device's rule, csrc/render.hip render_forward_auto_kernel: rows below 6.5 duplicates per
        # visible Gaussian; the A/B switches force one) and the lanes IT issued: 64 per wave step of the row mapping (counted with the
        # octagon-vs-cell culling the kernel uses), 64 per quadrant test otherwise.  K7 always walks quadrants.
        rows_run = True if args.row_mapped else (False if args.quadrant_mapped else 2 * D < 13 * V)
        fwd_lane_tests = (blend_counts["row_mapping_steps_octagon_culling"] if rows_run else blend_counts["quadrant_tests"]) * 64
        valu = {"peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "flop_per_pair": 260, **blend_counts, "lane_tests": lane_tests,
                "lane_utilisation": round(blend_counts["contributing_pairs"] / lane_tests, 4) if lane_tests else None,
                "forward_mapping": "16-lane rows on independent entries" if rows_run else "one entry on 64 lanes (8x8 quadrants)",
                "forward_lane_utilisation": round(blend_counts["contributing_pairs"] / fwd_lane_tests, 4) if fwd_lane_tests else None,
                "backward_lane_utilisation": round(blend_counts["contributing_pairs"] / (blend_counts["quadrant_tests_with_a_hit"] * 64), 4) if blend_counts["quadrant_tests_with_a_hit"] else None,
                "issued": None if issued is None else round(issued, 2), "issued_frac": None if issued is None else round(issued / FP32_VALU_PEAK_TFLOPS, 4),
                "useful": None if useful is None else round(useful, 2), "useful_frac": None if useful is None else round(useful / FP32_VALU_PEAK_TFLOPS, 4),
                "issue_roof": {"render_forward_kernel": valu_issue_roof("render_forward_kernel", group_ms["blend_fwd"], args),
                               "render_backward_kernel": valu_issue_roof("render_backward_kernel", group_ms["blend_bwd"], args, blend_counts["entries_with_a_hit"])},
                "note": "issued = 64 lanes x quadrant tests, useful = pairs that contribute.  Neither memory (traffic ~ algorithmic bytes at a fifth "
                        "of the HBM rate) nor the number of resident waves bounds the blend kernels (a fourth wave per SIMD, built for the two-pixel K7, "
                        "changes nothing): they run at issue_roof.*.datasheet_issue_frac of the rate the vector unit would reach if every instruction "
                        "were a two-cycle VOP2 and at issue_roof.*.fitted_issue_cost_frac of the rate their actual instruction forms allow (DPP, "
                        "transcendentals, lane swaps, three-source / SGPR-source VOP3 at their measured throughput costs) -- the second binds (DESIGN.md 4)",
                "D_eff_blend_bytes": blend_counts["staged_entries_D_eff"] * 296 + npx * 160}
        # The same fractions with the list entries the blend kernels actually stage (D_eff: what lies behind a tile's saturation
        # point is never loaded) in place of all D duplicates: the D-based figures count bytes no kernel touches -- a factor of two
        # on the 4K scene.
        d_eff = blend_counts["staged_entries_D_eff"]
        ab_eff = {"blend_fwd": d_eff * 76 + npx * 60, "blend_bwd": d_eff * 76 + d_eff * 72 * 2 + npx * (60 + 40)}
        frac_eff = ab_eff[dominant] / (group_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS if group_ms[dominant] else None
        blend_frac_eff = (ab_eff["blend_fwd"] + ab_eff["blend_bwd"]) / (blend_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if blend_ms else None
        # counter traffic of BOTH blend kernels against the algorithmic bytes -- the D-based figure of SURVEY 8(d) and the one computed on
        # the entries the kernels actually stage (D_eff): the second is the honest "how many bytes per byte needed" ratio
        traffic_ratios = {}
        for stage, kern in (("blend_fwd", "render_forward_kernel"), ("blend_bwd", "render_backward_kernel")):
            t, _ = pmc_traffic(kern, args)
            if t:
                traffic_ratios[kern] = {"traffic_bytes_per_launch": int(t), "traffic_over_D_bytes": round(t / ab[stage], 3),
                                        "traffic_over_D_eff_bytes": round(t / ab_eff[stage], 3)}
        out = {
            "metric": f"Msplats/s fwd+bwd @{W}x{H}, {P / 1e6:g}M Gaussians", "value": round(value, 3), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "timed_region_ms": round(elapsed * 1e3, 3),
            "step_ms_percentiles": {"p10": pct(0.1), "p50": pct(0.5), "p90": pct(0.9), "max": round(per_step[-1], 4)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.label, "baseline_config": args.tag,
                       "gaussians": P, "width": W, "height": H, "sh_degree": deg, "visible": V, "duplicates_D": D, "visible_per_P": round(V / P, 4), "D_per_P": round(D / P, 3),
                       "frames_per_step": world * K, "frames_per_rank": K,
                       "parallelism": (f"frame-sharded dp{world}" if multi else "single GPU") + (f", {K} frames accumulated per exchange" if K > 1 else ""),
                       "camera_rig": args.rig,
                       **({"gradient_exchange": ("all-gather of 12-B colour gradients (started between K7 and K8) + local SH expansion + "
                                                 "all-reduce of 40 B/Gaussian" if args.exchange == "factored" else
                                                 ("visibility bitmaps all-gathered (P / 8 B), then ONE all-reduce of the 232 B of every row reached on at least one rank"
                                                  if args.exchange == "compacted" else "all-reduce of 232 B/Gaussian")),
                           "exchange_rows_union_over_P": (round(timed_exchange.get("rows_union", P) / P, 4) if args.exchange == "compacted" else None),
                           "exchange_bytes_dense_allreduce_equivalent": P * 232,
                           "exchange_ms_per_step_rank0_host_wait": round(timed_exchange["ms"] / args.steps, 4),
                           "exchange_bytes_sent_per_step_rank0": int(timed_exchange["bytes"] / args.steps),
                           "exchange_early_starts_per_step": timed_exchange["early_starts"] / args.steps,
                           "exchange_selfcheck": exchange_check, "backend": dist.get_backend(),
                           "world_size_seen_by_each_rank": worlds_seen, "rccl": rccl_info,
                           "exchange": exchange_report(exchange_detail, P, world, K, ms_per_step, args.exchange)} if multi else {})},
            "roofline": {"bound": "hbm", "kernel": dom_kernel,
                         "achieved": None if ach is None else round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 5),
                         "frac_D_eff": None if frac_eff is None else round(frac_eff, 5),
                         "algorithmic_bytes_per_launch": ab[dominant], "algorithmic_bytes_per_launch_D_eff": ab_eff[dominant],
                         "avg_launch_ms": round(group_ms[dominant], 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_D_eff_bytes": traffic_ratios.get(dom_kernel, {}).get("traffic_over_D_eff_bytes"), "blend_traffic_ratios": traffic_ratios,
                         "measured_streaming_ceiling": ceiling, "hbm_bound_kernels": hbm_bound,
                         "note": "HBM fraction as BASELINE.json defines it; the blend kernels are instruction-issue / latency-bound, not HBM-bound (valu.issue_roof, DESIGN.md 4)",
                         "north_star_blend_fwd_bwd": {"algorithmic_bytes": ab["blend_fwd"] + ab["blend_bwd"], "ms": round(blend_ms, 4),
                                                      "achieved": None if blend_gbs is None else round(blend_gbs, 2),
                                                      "frac": None if blend_gbs is None else round(blend_gbs / HBM_PEAK_GBS, 5),
                                                      "frac_D_eff": None if blend_frac_eff is None else round(blend_frac_eff, 5)},
                         "valu": valu,
                         "whole_op": {"algorithmic_bytes": sum(ab.values()), "kernel_ms": round(kernels_ms, 4),
                                      "frac": round(sum(ab.values()) / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kernels_ms else None}},
            "stage_ms": {k: (None if v is None else round(v, 4)) for k, v in stage_ms.items()},
        }
        if world == 1 and not multi and not args.no_train_step:
            try:
                out["train_step"] = train_step_section(args, params, cam, dev, D, V)
            except Exception as e:   # an untimed extra must never cost the line
                out["train_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not multi and not args.no_train_step and args.tag == "c3":
            try:
                out["camera_inside_scene"] = camera_inside_scene_section(P, W, H, deg, dc, da, dev)
            except Exception as e:   # an untimed extra must never cost the line
                out["camera_inside_scene"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not multi and not args.no_train_step and args.tag == "c3":
            try:   # the reference's documented operating point (`-r 4`: 480x320 frames) and the forward's launch floor
                out["reference_operating_point_r4"] = reference_operating_point_section(deg, dev)
            except Exception as e:   # an untimed extra must never cost the line
                out["reference_operating_point_r4"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, g_cpu, cam, dc_cpu, da_cpu)
    # The JSON line is the LAST thing on stdout.  With NCCL_DEBUG set RCCL printf()s a version banner into libc's stdout buffer at init, which
    # a pipe only sees when the process exits -- i.e. after the line, and once per rank.  So: every rank flushes libc's buffers now, the ranks
    # meet, rank 0 prints, and everybody points fd 1 at /dev/null for whatever else gets flushed at exit.
    _flush_c_stdio()
    if multi:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    _discard_further_stdout()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
