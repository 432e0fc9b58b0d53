#!/bin/bash
# same-box A/B: split-K of the shallow few-tile lean GEMMs (UDT_LEAN_SPLITK=0) vs the default policy
mkdir -p gpurun_out/r03p
for v in 0 1 0 1; do
  UDT_INFLIGHT_LOCKSTEP=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-reference-default > gpurun_out/r03p/ab_lock_$v.json 2> gpurun_out/r03p/ab_lock_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03p/ab_lock_$v.json").read().strip().splitlines()[-1])
print("UDT_INFLIGHT_LOCKSTEP=$v value", round(d["value"],3), "one_batch", round(d["value_one_batch"],3), "unet_ms", round(d["unet_ms_per_sampler_step"],3), "gemm frac", round(d["roofline_classes"]["gemm"]["frac"],4))
PY
done
