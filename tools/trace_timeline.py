"""Print a window of a rocprofv3 --kernel-trace run as a timeline (start, end, duration in us, queue, stream, kernel):
   python tools/trace_timeline.py <results.db> [rows from the end = 140] [rows to print = 70]
   (how kernels of two streams overlap: the two-stream cnn plan of run_cnn)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 140
count = int(sys.argv[3]) if len(sys.argv) > 3 else 70
rows = list(db.execute("select name, start, end, queue_id, stream_id from kernels order by start"))


def short(n):
    m = re.search(r"kws::(\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:30]


i0 = max(0, len(rows) - back)
t0 = rows[i0][1]
for r in rows[i0:i0 + count]:
    print(f"{(r[1] - t0) / 1000:9.1f} {(r[2] - t0) / 1000:9.1f} {(r[2] - r[1]) / 1000:7.1f} q{r[3]} s{r[4]} {short(r[0])}")
