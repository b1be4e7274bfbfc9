# A/B timing of two builds ON THE SAME GPU BOX (boxes differ by up to 5 %): put the two libraries into ab/libA.so and ab/libB.so
# (ab/ is git-ignored but travels with gpurun), then: gpurun -- 'bash tools/ab_bench.sh [bench args]'
L=streetunveiler_amd/lib/libsurfel_raster.so
cp $L /tmp/lib_keep.so
for r in 1 2; do for v in ${AB_VARIANTS:-A B}; do
  cp ab/lib$v.so $L
  timeout 150 python bench.py --no-cpu-baseline --no-train-step "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], {k: (None if x is None else round(x, 4)) for k, x in d['stage_ms'].items()})"
done; done
cp /tmp/lib_keep.so $L
