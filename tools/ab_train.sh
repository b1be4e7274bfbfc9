# A/B of kernel builds on the fused training-step pattern, ON THE SAME GPU BOX: every ab/lib*.so (ab/ is git-ignored but travels with
# gpurun) runs tools/train_step_bench.py under rocprofv3 --kernel-trace --stats; prints ms per iteration and the top sr:: kernels.
# usage: gpurun -- 'bash tools/ab_train.sh'      (build variants with: SR_EXTRA_HIPCC_FLAGS="-D..." python -c "from streetunveiler_amd.build import build; build(libdir='ab/X')" && cp ab/X/libsurfel_raster.so ab/libX.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for lib in ab/lib*.so; do
  v=$(basename $lib .so); D=$R/gpurun_out/ab_train/$v; rm -rf $D; mkdir -p $D
  echo "== $v"
  SURFEL_RASTER_LIB=$R/$lib python tools/train_step_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1
  SURFEL_RASTER_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python tools/train_step_bench.py --steps 6 --warmup 2 > $D/log.txt 2>&1
  python tools/kstats.py $(find $D -name "*kernel_stats.csv" | head -1) | grep -E "class_|render_|preprocess_" | sort -t g -k2 | head -12
done
