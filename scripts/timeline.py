#!/usr/bin/env python3
"""Print the GPU kernel timeline (start offset, duration, gap before) of the last few train steps from a rocpd db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# find last occurrences of render_backward to delimit train steps
idx = [i for i, r in enumerate(rows) if "render_backward" in r[0]]
if len(idx) < 3:
    sys.exit("no train steps")
a, b = idx[-3], idx[-2]
# a step = from just after preprocess_backward of step k-1 to preprocess_backward of step k
seg = rows[a:b + 2]
t0 = seg[0][1]
prev_end = None
for n, s, e in seg:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f}us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {n[:70]}")
    prev_end = e
