# Counter comparison of library variants on one box: ab/lib<V>.so for V in $AB_VARIANTS; prints per-launch counter averages of one kernel.
# usage: gpurun -- 'AB_VARIANTS="A B" AB_KERNEL=render_backward bash tools/ab_pmc.sh'
L=streetunveiler_amd/lib/libsurfel_raster.so
cp $L /tmp/lib_keep.so
for v in ${AB_VARIANTS:-A B}; do
  cp ab/lib$v.so $L
  bash tools/pmc_pass.sh abpmc_$v p1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
  bash tools/pmc_pass.sh abpmc_$v p2 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_DATA_FIFO_FULL
  python - $v <<'P'
import csv, sys, glob, collections, re
v = sys.argv[1]
import os
k = os.environ.get("AB_KERNEL", "render_backward")
for f in sorted(glob.glob(f"gpurun_out/abpmc_{v}/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if re.search(k, r["Kernel_Name"]): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in acc.items()}, "(millions per launch)")
P
done
cp /tmp/lib_keep.so $L
