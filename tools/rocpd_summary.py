"""Dump the per-kernel summary (top_kernels view) of a rocprofv3 rocpd database as CSV."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("kernel,calls,total_us,avg_us,percent")
for name, calls, tot, avg, pct in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*$", "", short).replace("void ", "").strip()
    if short.startswith("at::native") or short.startswith("__amd"):
        short = short[:60]
    print(f'"{short}",{calls},{tot:.1f},{avg:.3f},{pct:.3f}')
