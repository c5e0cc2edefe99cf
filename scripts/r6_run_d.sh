O=gpurun_out/r6d; mkdir -p $O
python tests/tools/debug_nan.py "rot NaN" "scale NaN" 2>&1 | grep -v libdrm > $O/debug_nan.log; head -60 $O/debug_nan.log
python scripts/probe_balance.py --dump $O/probe_order_on.npz > $O/probe_order_on.json 2>$O/probe.err
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get('stages_ms',{})
print(f"{sys.argv[2]:60s} value {d['value']:8.1f} ms/step {d['ms_per_step']:.4f} fwd {d.get('forward_fps',0):8.1f}", d.get('streams',{}).get('every_view_bit_identical_to_its_run_alone'))
PY
}
for rep in 1 2; do
for v in "--streams 1" "--streams 2" "--streams 3" "--streams 2 --option speculative_forward=2" "--streams 4 --option speculative_forward=2"; do
  python bench.py --views 8 --gpus 1 --no-cpu-baseline --steps 100 --warmup 20 $v > $O/s.json 2>$O/s.err || tail -3 $O/s.err
  line $O/s.json "views8 $v"
done; done | tee $O/streams_summary.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d $O/trace -o t -- python bench.py --views 8 --gpus 1 --streams 2 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $O/trace.log 2>&1
python scripts/stream_overlap.py $O/trace/t_results.db > $O/stream_overlap_views8_streams2.txt 2>&1; cat $O/stream_overlap_views8_streams2.txt
rm -rf $O/trace
