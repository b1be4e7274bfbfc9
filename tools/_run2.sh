cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "65536" 2>&1 | tail -5
python tools/tile_sweep.py > gpurun_out/sweep4k.log 2>&1; tail -6 gpurun_out/sweep4k.log | cut -c1-250
