#!/usr/bin/env python3
"""One screen per directory of bench.py lines: value, forward rate, ms/step, quantiles, stage times, roofline, parity, cpu_baseline.
usage: python scripts/summarize_bench_dir.py <dir> [pytest log]   (prints; redirect to <dir>/summary.txt)"""
import glob, json, sys

d_ = sys.argv[1]
if len(sys.argv) > 2:
    print("GPU suite (" + sys.argv[2] + "):", open(sys.argv[2]).read().strip().splitlines()[-1])
for f in sorted(glob.glob(d_ + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); s = d.get("stages_ms", {})
        if "value" in d:
            print(f"{f.split('/')[-1]:60s} {d['value']:8.1f} {d['unit']} fwd {d.get('forward_fps', 0):8.1f} fps ms/step {d['ms_per_step']} q {d.get('step_ms_quantiles')} sources {d.get('library', {}).get('kernel_source_sha')}")
            print("      stages", s)
            h = d.get("timed_region_host_ms")
            if h:
                print("      timed region (host clock)", {k: h[k] for k in ("p50", "mean", "max", "slow_steps") if k in h})
            r = d.get("roofline")
            if r:
                print("      roofline", {k: r.get(k) for k in ("bound", "kernel", "frac", "valu_busy", "avg_launch_ms")}, "8d", r.get("hbm", {}).get("by_survey_8d_bytes"), "traffic", r.get("traffic"))
            for k in ("parity", "cpu_baseline", "camera_sequence", "speedup_vs_reference_on_this_gpu"):
                if k in d:
                    print("     ", k, d[k])
            if "reference_on_this_gpu" in d:
                print("      ref", {k: v for k, v in d["reference_on_this_gpu"].items() if k != "what"})
        else:
            print(f.split('/')[-1], {k: v for k, v in d.items() if 'ms' in k or 'share' in k})
    except Exception as e:
        print(f, "FAILED", e)
