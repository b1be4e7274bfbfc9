# Forward-only mode (SR_FLAG_FORWARD_ONLY) against the training forward at C3 / C5: ms per frame and stage, and K1's / K6's HBM traffic in
# either mode (separate --pmc FETCH_SIZE / WRITE_SIZE passes per mode, --kernel-trace only).   usage (GPU box): bash tools/profile_forward_only.sh r05
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/${TAG}_forward_only; rm -rf $D; mkdir -p $D; cd $R
python tools/fwd_only_time.py $D/times.json > $D/times.log 2>&1
for mode in training_forward forward_only; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $D -o ${mode}_${ctr} -- python tools/fwd_only_time.py --configs c3 --modes $mode > $D/${mode}_${ctr}.log 2>&1
  done
done
python - "$D" "$TAG" <<'P'
import collections, csv, json, sys
D, tag = sys.argv[1], sys.argv[2]
out = json.load(open(f"{D}/times.json"))
def per_kernel(fn, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == counter and "sr::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
traffic = {}
for mode in ("training_forward", "forward_only"):
    fe, wr = per_kernel(f"{D}/{mode}_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE"), per_kernel(f"{D}/{mode}_WRITE_SIZE_counter_collection.csv", "WRITE_SIZE")
    traffic[mode] = {k: {"FETCH_SIZE_KiB": round(fe[k]), "WRITE_SIZE_KiB": round(wr.get(k, 0)), "traffic_bytes_per_launch": int((2 * fe[k] + wr.get(k, 0)) * 1024)}
                     for k in fe if k in ("sr::preprocess_forward_kernel", "sr::render_forward_auto_kernel", "sr::render_forward_kernel", "sr::render_forward_rows_kernel")}
s = out["c3"]["scene"]; V, P = s["visible"], s["P"]
alg = {"training_forward": V * (12 + 8 + 16 + 4 + 192) + P * 16 + V * 87, "forward_only": V * (12 + 8 + 16 + 4 + 192) + P * 16 + V * (87 - 36)}
for mode in traffic:
    k1 = traffic[mode].get("sr::preprocess_forward_kernel")
    if k1:
        k1["algorithmic_bytes"] = alg[mode]; k1["traffic_over_algorithmic"] = round(k1["traffic_bytes_per_launch"] / alg[mode], 3)
out["c3"]["hbm_traffic"] = traffic
out["how"] = ("tools/profile_forward_only.sh: tools/fwd_only_time.py (HIP events, 20 frames; library stage timers, 10 frames) + rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
              "in separate passes per mode; traffic = (2 FETCH + WRITE) x 1024; K1's algorithmic bytes per SURVEY 8(d), minus the 36-B sh_jac row in forward-only mode")
json.dump(out, open(f"{D}/{tag}_forward_only.json", "w"), indent=1)
print(json.dumps({m: traffic[m].get("sr::preprocess_forward_kernel") for m in traffic}))
P
find $D -maxdepth 1 -name "*.csv" -delete
