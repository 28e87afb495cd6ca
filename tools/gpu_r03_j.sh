#!/bin/bash
# Round-3 GPU call J: config 5's launch time against the solver's iteration cap (the launch waits for the env whose solves run into the cap).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for regime in standing collapsing; do
  for mi in 150 100 60 40 25; do
    timeout 300 python $R/bench.py --config 5 --atlas-regime $regime --no-cpu --steps 150 --warmup 30 --max-iter $mi 2>$O/err.txt | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('config 5 %-10s max_iter %3d: %.2f M env-steps/s, kernel mean %.3f ms max %.3f ms, sweeps (last sub-step) mean %.1f max %d' % ('$regime', $mi, b['value']/1e6, b['roofline']['kernel_ms_mean'], b['roofline']['kernel_ms_max'], b['state_at_end']['solver_iters_mean'], b['state_at_end']['solver_iters_max']))" 2>&1 | tee -a $O/c5_max_iter.txt
  done
done
cd $R; ( timeout 300 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q ) 2>&1 | tail -3
