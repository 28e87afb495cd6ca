#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs35.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
cd /tmp
for rep in 1 2; do for v in "" "--atlas-regime collapsing"; do
  timeout 300 python $R/bench.py --config 5 --no-cpu $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s %6.2f M  kernel %.4f ms'%('$v', d['value']/1e6, d['roofline']['kernel_ms_mean']))" | tee -a $O/c5.txt
done; done
timeout 300 python $R/tools/diag_atlas_phases.py standing 2>&1 | grep -E "waves|per sweep" | tee -a $O/c5.txt
