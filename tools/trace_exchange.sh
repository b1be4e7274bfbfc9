# kernel trace of the frame-parallel bench path on a one-rank RCCL group (GPU box): bash tools/trace_exchange.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/quick_exchange
rm -rf $D; mkdir -p $D
cd $R
export SURFEL_EXCHANGE_SINGLE_RANK=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --gpus 1 --no-cpu-baseline --steps 20 --warmup 5 > $D/trace_bench.log 2>&1
python tools/kstats.py $(find $D -name "*kernel_stats.csv" | head -1)
