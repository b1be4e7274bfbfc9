set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r01b
mkdir -p $D
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $D/trace_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $D -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $D/sq.log 2>&1
ls -la $D | head -30
find $D -name "*.csv" | head -20
python bench.py 2>&1 | tail -1 > $D/bench.json
