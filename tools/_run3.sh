cd $GRAFT_REPO_ROOT
for c in c3 c2 c5; do bash tools/profile_round.sh r02 $c > gpurun_out/prof_$c.log 2>&1; tail -2 gpurun_out/prof_$c.log; done
python tools/parity_report.py --gaussians 500000 > gpurun_out/r02_parity.log 2>&1; tail -3 gpurun_out/r02_parity.log
python tools/clustered_scene.py > gpurun_out/r02_clustered.log 2>&1; tail -3 gpurun_out/r02_clustered.log
python tools/time_class_distortions.py > gpurun_out/r02_class_dist.log 2>&1; tail -3 gpurun_out/r02_class_dist.log
python tools/tile_sweep.py > gpurun_out/sweep4k.log 2>&1; python tools/tile_sweep.py 3000000 1920 1080 > gpurun_out/sweep1080.log 2>&1; tail -2 gpurun_out/sweep1080.log
tools/ubench/valu_issue_ubench > gpurun_out/valu_issue_ubench.txt 2>&1
python tools/time_semantic.py > gpurun_out/r02_time_semantic.log 2>&1; python tools/time_rgb_and_semantic.py > gpurun_out/r02_time_rgbsem.log 2>&1; python tools/time_mask.py > gpurun_out/r02_time_mask.log 2>&1; tail -2 gpurun_out/r02_time_semantic.log gpurun_out/r02_time_rgbsem.log gpurun_out/r02_time_mask.log
