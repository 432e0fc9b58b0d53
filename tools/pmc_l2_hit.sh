#!/bin/bash
# L2 hit rate per kernel over a 2-step batch of the bench workload (eager launches, one stream): two PMC passes (TCC_HIT_sum,
# TCC_MISS_sum; no tracing domains).   usage (GPU box): bash tools/pmc_l2_hit.sh <sha>  > profiles/rNN_l2_hit.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for c in TCC_HIT_sum TCC_MISS_sum; do
  rm -rf /tmp/l2_$c
  UDT_GRAPHS=0 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/l2_$c -o p -- python $R/tools/predict_once.py 2 > /dev/null 2>&1
done
python - "$1" <<'PY'
import csv, sys, collections, glob
def load(c):
    f = glob.glob(f"/tmp/l2_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = (k[:k.index(">(") + 1] if ">(" in k else k.split("(")[0])[:64]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
h, m = load("TCC_HIT_sum"), load("TCC_MISS_sum")
print(f"# HEAD {sys.argv[1]}")
print("# L2 (TCC) requests of 128 B per launch and hit rate, per kernel, over one 2-step batch (4 images, CFG pair per call)")
rows = []
for k in h:
    if k in m and h[k][0] == m[k][0]:
        n = h[k][0]; hh, mm = h[k][1] / n, m[k][1] / n
        rows.append(((hh + mm) * n, k, n, hh + mm, hh / max(hh + mm, 1)))
for tot, k, n, req, rate in sorted(rows, reverse=True)[:24]:
    print(f"{k:66s} launches {n:5d}   requests/launch {req / 1e6:8.2f} M ({req * 128 / 1e6:8.1f} MB)   hit rate {rate * 100:5.1f} %")
PY
