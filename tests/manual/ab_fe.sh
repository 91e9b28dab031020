# Same-box A/B of the front-end leg for library variants: gpurun -- 'bash tests/manual/ab_fe.sh tag libA.so libB.so ...'
TAG=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    VINS_AB_LIB=$lib VINS_FE_DISTINCT=${VINS_FE_DISTINCT:-256} python tests/manual/gpu_fe_leg.py 40 2> gpurun_out/${TAG}_${lib%.so}_${rep}.err | tail -1 > gpurun_out/${TAG}_${lib%.so}_${rep}.json
    echo "$lib $rep $(cut -c1-400 gpurun_out/${TAG}_${lib%.so}_${rep}.json)"
  done
done
