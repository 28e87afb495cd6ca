#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_gpu_configs35.py -m gpu -q -k "valley or two_contacts or heightmap or ridge or height_map or config3 or sampled or per_env" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log | cut -c1-250
AB_ARGS="--config 3" bash tools/ab.sh 3 librsb.DRSB_X_BASE.so librsb.so | tee $O/ab_c3.txt
AB_ARGS="--config 2" bash tools/ab.sh 3 librsb.DRSB_X_BASE.so librsb.so | tee $O/ab_c2.txt
