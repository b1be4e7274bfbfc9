# rocprofv3 evidence for one bench.py workload (GPU box).  usage: bash tools/profile_round.sh <round tag, e.g. r02> <config c2|c3|c5> [suffix "extra bench.py args"]
# (suffix + extra args: a built VARIANT beside the shipped default, e.g. `r06 c3 coop "--backward-kernel coop"` -> r06_c3coop_*)
# Counters in their own passes with --kernel-trace only (never together with sys / hip / hsa traces).
set -e
TAG=$1; CFG=${2:-c3}; SUF=${3:-}; EXTRA=${4:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/${TAG}_${CFG}${SUF}
mkdir -p $D
cd $R
B="python bench.py --config $CFG --no-cpu-baseline --no-train-step $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- $B --steps 20 --warmup 5 > $D/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- $B --steps 3 --warmup 1 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- $B --steps 3 --warmup 1 > $D/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $D -o sq -- $B --steps 3 --warmup 1 > $D/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $D -o sq2 -- $B --steps 3 --warmup 1 > $D/sq2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $D -o sq3 -- $B --steps 3 --warmup 1 > $D/sq3.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o grbm -- $B --steps 3 --warmup 1 > $D/grbm.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $D -o l2 -- $B --steps 3 --warmup 1 > $D/l2.log 2>&1 || true
# counters first, then the plain bench run: bench.py quotes the newest profiles/<round>_<cfg>_{hbm_traffic,sq_counters}.json whose source digest is the
# current one, so the fresh counter files go into the box's profiles/ BEFORE the run whose line is kept (it used to quote the previous profile)
python tools/collect_profiles.py $D ${TAG}_${CFG}${SUF} $D/out
if [ -z "$SUF" ]; then cp $D/out/${TAG}_${CFG}_hbm_traffic.json $D/out/${TAG}_${CFG}_sq_counters.json profiles/; fi
python bench.py --config $CFG $EXTRA 2>$D/bench.err | tail -1 > $D/bench.json
python tools/collect_profiles.py $D ${TAG}_${CFG}${SUF} $D/out
ls $D/out
# the raw CSVs stay on the box (gpurun merges at most 64 MiB back): the condensed files in out/ and the logs are what is kept
find $D -maxdepth 1 -name "*.csv" -delete
