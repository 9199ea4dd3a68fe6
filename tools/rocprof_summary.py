"""Dump the per-kernel summary (top_kernels view) of a rocprofv3 rocpd .db into CSV (for profiles/)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("kernel,calls,total_us,avg_us,percent")
for n, c, t, a, p in rows:
    print('"%s",%d,%.3f,%.3f,%.3f' % (n.replace('"', "'"), c, t, a, p))
