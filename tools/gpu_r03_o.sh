#!/bin/bash
# Anderson step in the large-model classes: Atlas parity / population tests, config-5 bench with and without it, max_iter variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs35.py tests/test_gpu_kat.py tests/test_gpu_properties.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in "" "--anderson 0" "--max-iter 64" "--anderson 0 --max-iter 64" "--atlas-regime collapsing" "--atlas-regime collapsing --anderson 0"; do
  timeout 300 python $R/bench.py --config 5 --no-cpu $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-50s %6.2f M  ms/step %.4f kernel %.4f ms (max %.3f)'%('$v', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['kernel_ms_max']))" | tee -a $O/c5_ab.txt
done; done
timeout 300 python $R/tools/diag_atlas_phases.py standing > $O/diag_atlas.txt 2>&1
timeout 300 python $R/tools/diag_atlas_phases.py collapsing >> $O/diag_atlas.txt 2>&1
cat $O/diag_atlas.txt
timeout 300 python $R/bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config 2 %6.2f M kernel %.4f ms'%(d['value']/1e6, d['roofline']['kernel_ms_mean']))" | tee -a $O/c5_ab.txt
