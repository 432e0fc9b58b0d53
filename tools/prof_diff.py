"""steady-state per-evaluation kernel split: (stats of R2 repetitions - stats of R1 repetitions) / (R2 - R1); usage: prof_diff.py a.csv R1 b.csv R2"""
import csv, sys
def rd(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["kernel"]] = (int(r["calls"]), float(r["total_us"]))
    return d
a, r1, b, r2 = rd(sys.argv[1]), int(sys.argv[2]), rd(sys.argv[3]), int(sys.argv[4])
rows = []
for k, (c2, t2) in b.items():
    c1, t1 = a.get(k, (0, 0.0))
    n = (c2 - c1) / (r2 - r1); t = (t2 - t1) / (r2 - r1)
    if n > 0:
        rows.append((t, n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"per evaluation: {tot / 1e3:.2f} ms of kernel time, {sum(r[1] for r in rows):.0f} launches")
for t, n, k in rows[:45]:
    print(f"{t / 1e3:8.3f} ms  x{n:6.1f}  {t / n:8.1f} us  {k[:110]}")
