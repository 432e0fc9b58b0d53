#!/bin/bash
# same-box A/B of the bench line: bf16 (config #2) vs --fp8 (config #5's arithmetic on one GPU) vs --fp8 with bf16 attention.  usage: tools/ab_fp8.sh <out-dir> [extra bench flags]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/${1:-ab_fp8}; shift
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-extra-configs --no-reference-default "$@" 2>/dev/null | tail -1 > $O/bench_bf16.json
python bench.py --fp8 --no-cpu-baseline --no-extra-configs --no-reference-default "$@" 2>/dev/null | tail -1 > $O/bench_fp8.json
UDT_FP8_ATTN=0 python bench.py --fp8 --no-cpu-baseline --no-extra-configs --no-reference-default "$@" 2>/dev/null | tail -1 > $O/bench_fp8_linears_only.json
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
v = {}
for n in ("bf16", "fp8", "fp8_linears_only"):
    d = json.load(open(f"{o}/bench_{n}.json"))
    v[n] = d["value"]
    cls = {k: round(x["frac"], 3) for k, x in (d.get("roofline_classes") or {}).items() if isinstance(x, dict) and "frac" in x}
    print(f"{n:17s} {d['value']:.3f} images/s  {d['ms_per_step']:.1f} ms per batch  roofline {round(d['roofline']['frac'], 3)} classes {cls}  modes {d.get('images_per_s_by_launch_mode')}  unet ms {d.get('unet_ms_per_step')}")
print(f"fp8 (MX8 linears + e4m3 attention) / bf16 = {v['fp8'] / v['bf16']:.4f}   MX8 linears only / bf16 = {v['fp8_linears_only'] / v['bf16']:.4f}")
PY
