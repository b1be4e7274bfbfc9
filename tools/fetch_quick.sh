for c in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do n=$(echo $c | tr ' ' '_'); ( bash tools/pmc_pass.sh k6new $n $c ) || echo fail; done
python - <<'P'
import csv, glob, collections, re
for f in sorted(glob.glob("gpurun_out/k6new/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if re.search("render_forward|render_backward|preprocess_", k): acc[k.split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in d.items()})
P
