#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) output: per-kernel time stats (--kernel-trace) and per-kernel PMC sums (--pmc).
usage: rocpd_summary.py results.db [> summary.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    if "name" in cols and "duration" in cols:
        q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
             "group by name order by sum(duration) desc")
        rows = list(db.execute(q))
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for n, c, s, a, mn, mx in rows:
            print(f"{n[:90]:90s} {c:6d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
    try:
        ccols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        if ccols:
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            q = (f"select {kn}, counter_name, count(distinct dispatch_id), sum(value) / count(distinct dispatch_id), avg(duration) "
                 f"from counters_collection group by {kn}, counter_name order by {kn}")
            rows = list(db.execute(q))
            if rows:
                print("\nPMC counters (value summed over all XCDs/SEs, averaged per dispatch):")
                for n, cn, c, a, d in rows:
                    print(f"{n[:60]:60s} {cn:24s} dispatches={c:4d} per_dispatch={a:18.0f} avg_dur_us={d/1e3:9.1f}")
    except sqlite3.Error as e:
        print("no counters:", e)


if __name__ == "__main__":
    main(sys.argv[1])
