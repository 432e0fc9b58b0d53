"""per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()[:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if not any(s in k for s in ("lconv3", "lgemm", "conv3p", "gemm8", "attn")): continue
    print(k)
    for c in sorted(acc[k]): print(f"    {c:32s} {acc[k][c] / n[k][c]:16.1f}   (x{n[k][c]})")
