# kernel trace of a short bench run (GPU box): bash tools/quick_trace.sh <config> [extra bench args]; prints the per-kernel stats of everything but the blend kernels' neighbours
set -e
CFG=${1:-c3}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/quick_$CFG
rm -rf $D; mkdir -p $D
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --config $CFG --no-cpu-baseline --steps 20 --warmup 5 "$@" > $D/trace_bench.log 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
cut -c1-140 $F | head -24
