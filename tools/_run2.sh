cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_render_api.py -m gpu -x -q -k class_dist 2>&1 | tail -15
python tools/time_class_distortions.py 2>&1 | tail -4
