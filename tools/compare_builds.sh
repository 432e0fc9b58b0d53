#!/bin/bash
# usage (on the GPU box): tools/compare_builds.sh dirA dirB ...   -> one line per run: images/s, ms/batch, UNet ms/step, conv TF/s, conv us/launch
cd "$(dirname "$0")/.."
for rep in 1 2; do
for d in "$@"; do
  (cd "$d" && python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', round(d['value'],3), round(d['ms_per_step'],1), round(d['unet_ms_per_sampler_step'],2), round(d['roofline']['achieved'],1), round(d['roofline']['avg_launch_us'],1))")
done; done
