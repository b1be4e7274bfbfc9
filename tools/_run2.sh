set -x
cd $GRAFT_REPO_ROOT
python tools/parity_report.py --gaussians 500000 2>&1 | grep "vs F64\|config" | tail -30
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
