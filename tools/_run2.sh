set -x
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/strict_report.jsonl
SR_PARITY_REPORT=gpurun_out/strict_report.jsonl python -m pytest tests/test_gpu_strict_parity.py -m gpu -q 2>&1 | tail -15
cat gpurun_out/strict_report.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
