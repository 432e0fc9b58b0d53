#!/bin/bash
# same-box A/B of one environment switch on the bench: tools/ab_bench_env.sh VAR valueA valueB   (runs A B A B)
# e.g. tools/ab_bench_env.sh UDT_GN_EPI 1 0   -> gpurun_out/ab_<VAR>.txt
VAR=$1; A=$2; B=$3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
for v in $A $B $A $B; do
  env $VAR=$v timeout 600 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-reference-default --no-extra-configs > /tmp/ab_$VAR.json 2> /tmp/ab_$VAR.err
  python - <<PY
import json
d=json.loads(open("/tmp/ab_$VAR.json").read().strip().splitlines()[-1])
print("$VAR=$v value", round(d["value"],3), "one_batch", round(d["value_one_batch"],3), "unet_ms", round(d["unet_ms_per_sampler_step"],3), "gemm frac", round(d["roofline_classes"]["gemm"]["frac"],4))
PY
done | tee $R/gpurun_out/ab_$VAR.txt
