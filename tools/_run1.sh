set -x
export SR_HEAD=$(cat .sr_head 2>/dev/null)
bash tools/profile_round.sh r06 c3 coop "--backward-kernel coop" > gpurun_out/prof_c3coop.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_trace -o t -- python tools/small_frame_trace.py --capacity > gpurun_out/small_trace/run.log 2>&1
python tools/small_frame_trace.py > gpurun_out/small_trace/default.log 2>&1
python tools/small_frame_trace.py --capacity > gpurun_out/small_trace/capacity.log 2>&1
find gpurun_out/small_trace -name "*kernel_trace.csv" -delete
ls gpurun_out/small_trace
