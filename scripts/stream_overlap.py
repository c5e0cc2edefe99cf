#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace rocpd db of `bench.py --views V --streams S`: which kernels ran TOGETHER with a render kernel of another
stream (queue), and for how long -- the evidence that independent views overlap on one GPU.
usage: stream_overlap.py results.db [> summary.txt]"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
    if qcol is None:
        sys.exit(f"no queue / stream column in the kernels view: {cols}")
    rows = list(db.execute(f"select name, start, end, {qcol} from kernels order by start"))
    wg = [r for r in rows if "wg::" in r[0]]
    if not wg:
        sys.exit("no wg:: kernels in the trace")
    # the last 40 % of the trace: the timed regions
    t_lo = wg[0][1] + int(0.6 * (wg[-1][2] - wg[0][1]))
    wg = [r for r in wg if r[1] >= t_lo]
    queues = sorted({r[3] for r in wg})
    span = wg[-1][2] - wg[0][1]
    busy = sum(r[2] - r[1] for r in wg)
    print(f"kernels {len(wg)}, queues {queues}, window {span / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us "
          f"(= {busy / span:.3f} x the window: above 1 means kernels of different queues ran side by side)")
    import re

    def short(n):   # "void wg::render_forward_kernel<true>(int, ...)" -> "render_forward_kernel"
        m = re.search(r"wg::(\w+)", n)
        return (m.group(1) if m else n)[:34]
    render = [r for r in wg if "render_forward" in r[0] or "render_backward" in r[0]]
    together = defaultdict(float)
    for n, s, e, q in render:
        for n2, s2, e2, q2 in wg:
            if q2 == q or e2 <= s or s2 >= e:
                continue
            together[(short(n), short(n2))] += (min(e, e2) - max(s, s2)) / 1e3
    tot = defaultdict(float)
    for n, s, e, q in render:
        tot[short(n)] += (e - s) / 1e3
    print(f"{'render kernel':28s} {'resident beside it (other queue)':34s} {'us together':>12s} {'share of the render kernel':>28s}")
    for (a, b), us in sorted(together.items(), key=lambda kv: -kv[1])[:16]:
        print(f"{a:28s} {b:34s} {us:12.1f} {us / tot[a]:28.3f}")
    # one example: the first render_backward of the window and what overlapped it
    for n, s, e, q in render:
        if "render_backward" in n:
            print(f"\nexample: {short(n)} on queue {q}, {(e - s) / 1e3:.1f} us; kernels of other queues inside its span:")
            for n2, s2, e2, q2 in wg:
                if q2 != q and e2 > s and s2 < e:
                    print(f"   +{(s2 - s) / 1e3:8.1f} us  dur {(e2 - s2) / 1e3:8.1f}  queue {q2}  {short(n2)}")
            break


if __name__ == "__main__":
    main(sys.argv[1])
