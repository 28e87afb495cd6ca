#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
timeout 300 python tools/diag_sampled.py > $O/diag_sampled.txt 2>&1; grep spacing $O/diag_sampled.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
cd /tmp
for c in 2 3; do for rep in 1 2; do timeout 300 python $R/bench.py --no-cpu --config $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $c %6.2f M kernel %.4f ms'%(d['value']/1e6, d['roofline']['kernel_ms_mean']))" | tee -a $O/bench.txt; done; done
