# rocprofv3 evidence for one bench.py workload (GPU box).  usage: bash tools/profile_round.sh <round tag, e.g. r02> <config c2|c3|c5>
# Counters in their own passes with --kernel-trace only (never together with sys / hip / hsa traces).
set -e
TAG=$1; CFG=${2:-c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/${TAG}_${CFG}
mkdir -p $D
cd $R
B="python bench.py --config $CFG --no-cpu-baseline --no-train-step"
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- $B --steps 20 --warmup 5 > $D/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- $B --steps 3 --warmup 1 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- $B --steps 3 --warmup 1 > $D/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $D -o sq -- $B --steps 3 --warmup 1 > $D/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $D -o sq2 -- $B --steps 3 --warmup 1 > $D/sq2.log 2>&1
python bench.py --config $CFG 2>$D/bench.err | tail -1 > $D/bench.json
python tools/collect_profiles.py $D ${TAG}_${CFG} $D/out
ls $D/out
