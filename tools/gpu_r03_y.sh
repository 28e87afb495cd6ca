#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
SWEEP="b200cbf HEAD b200cbf HEAD b200cbf HEAD" bash tools/gpu_r03_x.sh
cd /tmp
for c in 3; do for rep in 1 2; do python $R/bench.py --no-cpu --config $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $c %6.2f M kernel %.4f ms'%(d['value']/1e6, d['roofline']['kernel_ms_mean']))"; done; done
