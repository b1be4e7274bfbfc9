# rocprofv3 evidence for the training-step pattern (tools/train_step_bench.py: render_train_view, one plan) at the C3 size: kernel stats, HBM traffic, SQ counters.
# usage (GPU box): bash tools/profile_train_step.sh <round tag, e.g. r05>      -> gpurun_out/<tag>_train_step/out/<tag>_train_step_*.{csv,json}
# Counters in their own passes with --kernel-trace only (never together with sys / hip / hsa traces).
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/${TAG}_train_step
rm -rf $D; mkdir -p $D
cd $R
B="python tools/train_step_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- $B --steps 10 --warmup 2 > $D/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- $B --steps 2 --warmup 1 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- $B --steps 2 --warmup 1 > $D/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $D -o sq -- $B --steps 2 --warmup 1 > $D/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $D -o sq2 -- $B --steps 2 --warmup 1 > $D/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o grbm -- $B --steps 2 --warmup 1 > $D/grbm.log 2>&1
python tools/train_step_bench.py --steps 10 --warmup 3 2>$D/bench.err | tail -1 > $D/bench.json
python tools/collect_profiles.py $D ${TAG}_train_step $D/out
ls $D/out
# the raw CSVs stay on the box (gpurun merges at most 64 MiB back): the condensed files in out/ and the logs are what is kept
find $D -maxdepth 1 -name "*.csv" -delete
