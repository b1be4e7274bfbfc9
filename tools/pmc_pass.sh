# One rocprofv3 --pmc pass around a short bench.py run; usage: tools/pmc_pass.sh <outdir> <name> <counters...> [-- bench args]
# (counters in their own run, kernel-trace only -- never combined with sys/hip/hsa traces)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$1; NAME=$2; shift 2
CTRS=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do CTRS="$CTRS $1"; shift; done
[ "$1" = "--" ] && shift
mkdir -p $D
cd $R
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $D -o $NAME -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step "$@" > $D/$NAME.log 2>&1
