#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
SWEEP="base HEAD base HEAD base HEAD" bash tools/gpu_r03_x.sh
cd /tmp
for rep in 1 2; do for l in base HEAD; do if [ $l = HEAD ]; then L=$R/raisimlib_amd/lib/librsb.so; else L=$R/raisimlib_amd/lib/librsb.$l.so; fi; RSB_LIB_PATH=$L python $R/bench.py --no-cpu --config 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config 3 $l %6.2f M kernel %.4f ms'%(d['value']/1e6, d['roofline']['kernel_ms_mean']))"; done; done
