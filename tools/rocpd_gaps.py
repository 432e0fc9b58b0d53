"""Busy time vs idle gaps of the kernel stream in a rocprofv3 rocpd database (last `frac` of the run)."""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = list(db.execute(f"select name, start, end from {view} order by start"))
n0 = int(len(rows) * (1 - frac))
rows = rows[n0:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)]
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f}%)  gaps {sum(gaps)/1e6:.3f} ms  mean gap {sum(gaps)/len(gaps)/1e3:.2f} us")
h = collections.Counter()
for g in gaps:
    h[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count):", dict(sorted(h.items())))
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*$", "", short).replace("void ", "").strip()[:70]
    agg[short][0] += 1; agg[short][1] += (e - s)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:9.3f} ms  x{n:5d}  avg {t/n/1e3:7.2f} us  {k}")
