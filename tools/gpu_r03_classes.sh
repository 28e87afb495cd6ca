#!/bin/bash
# what the opt-in kernel classes cost: config 3 with a second contact per primitive (least angle between the normals 26 / 45 / 60 deg),
# configs 2 / 3 with the trapezoid / explicit Euler scheme
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03classes; mkdir -p $O; rm -f $O/classes.txt; cd $R
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -q -k "two_contacts or valley" ) 2>&1 | tail -2
cd /tmp
run() { timeout 300 python $R/bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['state_at_end']; print('%-50s %7.2f M env-steps/s, kernel %.4f ms, contacts/env %.2f, sweeps mean %.2f max %d, resets/step %.1f'%('$*', d['value']/1e6, d['roofline']['kernel_ms_mean'], s['contacts_per_env'], s['solver_iters_mean'], s['solver_iters_max'], d['config']['regime']['resets_per_control_step_mean']))" | tee -a $O/classes.txt; }
run --config 3
run --config 3 --hm-contacts 2
run --config 3 --hm-contacts 2 --hm-angle 60
run --config 3 --hm-contacts 2 --hm-angle 26
run --config 3 --integration trapezoid
run --config 2
run --config 2 --integration trapezoid
run --config 2 --integration euler
