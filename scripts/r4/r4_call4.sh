#!/bin/bash
# round 4, GPU call 4: the driver's own command line (20 steps, 5 warm-up) -- where is the mean-vs-p50 gap?
O=gpurun_out/r4c4; mkdir -p $O
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.$rep.json 2> $O/driver_cmd.$rep.err
done
python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/steps200_warm5.json 2> $O/steps200_warm5.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value", d["value"], "ms_per_step", d["ms_per_step"], "quantiles", d.get("step_ms_quantiles"))
        print("   host:", d.get("timed_region_host_ms"))
        print("   camera_sequence:", d.get("camera_sequence"))
        print("   parity vs ref:", d.get("reference_on_this_gpu",{}).get("product_vs_reference"))
        print("   spec:", d["library"]["speculative_forward"])
    except Exception as e: print(f, "FAILED", e)
PY
tail -3 $O/driver_cmd.1.err
