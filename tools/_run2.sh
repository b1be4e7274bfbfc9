cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=8 2>&1 | tail -20
