#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c55; mkdir -p $O
timeout 400 python scripts/bench_wildgaussians_step.py --real-caller --optins --tall-linear --steps 10 --warmup 3 2>&1 | tail -1 > $O/real_optins_tall.json; cut -c1-400 $O/real_optins_tall.json; python -c "
import json; d=json.load(open('$O/real_optins_tall.json')); print(d['train_step_ms'], d['loss_first'], d['loss_last'])"
