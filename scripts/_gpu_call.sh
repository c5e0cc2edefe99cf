cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c26; mkdir -p $O
timeout 900 python tests/tools/stress_sweep_vs_reference.py 30000 30000 > $O/stress_sweep_vs_reference_30k.txt 2>&1; tail -1 $O/stress_sweep_vs_reference_30k.txt | cut -c1-600
timeout 600 python tests/tools/stress_sweep.py 40000 2000 > $O/stress_sweep_vs_oracle.txt 2>&1; tail -2 $O/stress_sweep_vs_oracle.txt | cut -c1-400
