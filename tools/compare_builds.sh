#!/bin/bash
# usage (on the GPU box): tools/compare_builds.sh spec spec ...   with spec = dir[:ENV=VAL[,ENV=VAL]]
# Same-box A/B (box-to-box variation on the pool is ~10 %): every spec is benchmarked REPS times, interleaved; one line
# (extra bench flags: BENCH_ARGS="--in-flight 4");  per run: images/s (on-config value), ms per step, UNet ms per sampler step, conv TF/s, conv us/launch, GEMM TF/s
cd "$(dirname "$0")/.."
REPS=${REPS:-2}
for rep in $(seq $REPS); do
for spec in "$@"; do
  d=${spec%%:*}; envs=""
  [ "$spec" != "$d" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  (cd "$d" && env $envs python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-configs --no-mode-table $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
g=d.get('roofline_classes',{}).get('gemm',{}).get('achieved',0)
print('$spec'.ljust(28), 'img/s %.3f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'unet_ms %.2f' % d['unet_ms_per_sampler_step'], 'conv %.0f TF %.1f us' % (d['roofline']['achieved'], d['roofline']['avg_launch_us']), 'gemm %.0f TF' % g)")
done; done
