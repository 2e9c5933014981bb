"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd SQLite file as text."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
print(f"# rocprofv3 --kernel-trace --stats summary of {db.split('/')[-1]} (durations in us)")
print(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, total, avg, pct in cur.fetchall():
    print(f"{calls:>6} {total:>14.1f} {avg:>12.2f} {pct:>7.2f}  {name}")
try:
    cur = c.execute("select name, count(*), avg(end-start)/1000.0 from kernels group by name")
except Exception:
    pass
