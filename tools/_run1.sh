set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a_bench_c3.json 2> gpurun_out/r02a_bench_c3.err
tail -c 600 gpurun_out/r02a_bench_c3.json
bash tools/pmc_pass.sh r02a sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
bash tools/pmc_pass.sh r02a sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_LDS
bash tools/pmc_pass.sh r02a sq3 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_INSTS_VMEM
ls gpurun_out/r02a
