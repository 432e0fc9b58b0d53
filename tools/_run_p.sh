#!/bin/bash
# same-box A/B: GroupNorm statistics from the lean epilogues (UDT_GN_EPI=1) vs gn_stats kernels (0)
mkdir -p gpurun_out/r03p
for v in 1 0 1 0; do
  UDT_GN_EPI=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-reference-default > gpurun_out/r03p/ab_gnepi_$v.json 2> gpurun_out/r03p/ab_gnepi_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03p/ab_gnepi_$v.json").read().strip().splitlines()[-1])
print("UDT_GN_EPI=$v value", round(d["value"],3), "one_batch", round(d["value_one_batch"],3), "unet_ms", round(d["unet_ms_per_sampler_step"],3))
PY
done
