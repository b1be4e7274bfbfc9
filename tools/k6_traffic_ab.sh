# K6 memory-side traffic of the two forward mappings on one box (round-4 review item 3): FETCH_SIZE / L2 hit counters per mapping.
# usage: gpurun -- 'bash tools/k6_traffic_ab.sh'
for m in row quadrant; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '_')
    ( bash tools/pmc_pass.sh k6ab_$m $n $c -- --$m-mapped ) || echo "pass $m $n failed"
  done
done
python - <<'P'
import csv, glob, collections, re
for m in ("row", "quadrant"):
    for f in sorted(glob.glob(f"gpurun_out/k6ab_{m}/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if re.search("render_forward|render_backward|preprocess_", k): acc[k.split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(m, k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in d.items()}, "(millions per launch)")
P
