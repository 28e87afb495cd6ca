#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
if [ -f raisimlib_amd/lib/librsb.base.so ]; then SWEEP="base HEAD base HEAD base HEAD" bash tools/gpu_r03_x.sh; fi
