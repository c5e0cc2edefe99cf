#!/bin/bash
# On the GPU box: the PMC byte counters against kernels of known HBM bytes (scripts/pmc_calibration.hip) -> gpurun_out/<tag>/pmc_calibration.json
# usage: scripts/r6/pmc_calibrate.sh <tag>     (the binary is built where hipcc is: wild-gaussians_amd/build/pmc_calibration)
set -u
TAG=${1:-r6_cal}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
BIN=wild-gaussians_amd/build/pmc_calibration
$BIN > $O/known_bytes.json
rocprofv3 --pmc FETCH_SIZE -d $O/f -o p -- $BIN > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o p -- $BIN > $O/w.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/t -o p -- $BIN > $O/t.log 2>&1
python scripts/rocpd_summary.py $O/f/p_results.db > $O/fetch_summary.txt 2>&1
python scripts/rocpd_summary.py $O/w/p_results.db > $O/write_summary.txt 2>&1
python scripts/rocpd_summary.py $O/t/p_results.db > $O/kernel_trace_summary.txt 2>&1
rm -rf $O/f $O/w $O/t
python - $O <<'PY'
import json, re, sys
O = sys.argv[1]
known = json.load(open(f"{O}/known_bytes.json"))
def counters(path, name):
    out = {}
    for line in open(path):
        m = re.match(rf"(\S.*?)\s+{name}\s+dispatches=\s*(\d+)\s+per_dispatch=\s*(\d+)\s+avg_dur_us=\s*([\d.]+)", line)
        if m:
            k = re.sub(r"^void ", "", m.group(1)).split("(")[0]
            out[k] = (int(m.group(3)), float(m.group(4)))
    return out
f, w = counters(f"{O}/fetch_summary.txt", "FETCH_SIZE"), counters(f"{O}/write_summary.txt", "WRITE_SIZE")
res = {"what": "known HBM bytes of scripts/pmc_calibration.hip's kernels (every datum once, arrays of 576 - 768 MiB: past the 256 MiB Infinity Cache) over the rocprofv3 "
               "counters of the same launches; factor = known bytes / (counter KiB x 1024): what a counter has to be multiplied by for this access pattern",
       "kernels": {}}
for k, kb in known.items():
    row = dict(kb)
    if k in f:
        row["FETCH_SIZE_KiB"], row["avg_us_under_pmc"] = f[k]
        if kb["read_bytes"]:
            row["fetch_factor"] = round(kb["read_bytes"] / (f[k][0] * 1024.0), 4)
    if k in w:
        row["WRITE_SIZE_KiB"] = w[k][0]
        if kb["write_bytes"]:
            row["write_factor"] = round(kb["write_bytes"] / (w[k][0] * 1024.0), 4)
    res["kernels"][k] = row
json.dump(res, open(f"{O}/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
