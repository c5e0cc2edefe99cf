#!/usr/bin/env python3
"""Post-processor of a rocprofv3 --kernel-trace of a long fwd+bwd loop (scripts/diag_step_blips.py): is a slow step a GAP on the GPU's
timeline (the GPU starved: nothing to run for milliseconds -- the host side was late) or a STRETCHED kernel (the GPU paused or slowed with
work in hand)?  Lists every idle gap between consecutive kernels above `gap_ms` with the kernels either side, and every kernel that ran
longer than `stretch` x its own median.    usage: diag_stall_trace.py results.db [gap_ms=0.5] [stretch=2.5]"""
import json, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
gap_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stretch = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp") if "start_timestamp" in cols else (None, None)
if s is None:
    print(json.dumps({"error": "no start / end columns", "columns": cols}))
    sys.exit(1)
rows = list(db.execute(f"select name, {s}, {e} from kernels order by {s}"))
short = lambda n: n.split("(")[0].replace("void ", "")[:60]
by = {}
for n, a, b in rows:
    by.setdefault(n, []).append(b - a)
med = {n: sorted(v)[len(v) // 2] for n, v in by.items()}
gaps, stretched = [], []
end_so_far = rows[0][2]
for i, (n, a, b) in enumerate(rows):
    if i and a - end_so_far > gap_ms * 1e6:
        gaps.append(dict(at_kernel=i, idle_ms=round((a - end_so_far) / 1e6, 3), before=short(rows[i - 1][0]), after=short(n)))
    if b - a > stretch * med[n] and b - a > 0.2e6:
        stretched.append(dict(at_kernel=i, kernel=short(n), ms=round((b - a) / 1e6, 3), median_ms=round(med[n] / 1e6, 3)))
    end_so_far = max(end_so_far, b)
span = (rows[-1][2] - rows[0][1]) / 1e6
busy = sum(b - a for _, a, b in rows) / 1e6
print(json.dumps({"kernels": len(rows), "span_ms": round(span, 1), "sum_of_kernel_ms": round(busy, 1), "idle_gaps_over_%.2f_ms" % gap_ms: gaps[:60],
                  "n_gaps": len(gaps), "kernels_over_%.1fx_their_median" % stretch: stretched[:60], "n_stretched": len(stretched)}))
