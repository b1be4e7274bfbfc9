cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in "--gaussians 3000000" "--gaussians 500000 --no-aux" "--gaussians 6000000 --width 3840 --height 2160 --steps 10"; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/tmp_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp_bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done
