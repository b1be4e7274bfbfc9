cd $GRAFT_REPO_ROOT
for c in c3 c2 c5; do bash tools/profile_round.sh r02 $c > gpurun_out/prof_$c.log 2>&1; tail -3 gpurun_out/prof_$c.log; done
