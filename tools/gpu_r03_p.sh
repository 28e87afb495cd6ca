#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 300 python tools/diag_sampled.py > $O/diag_sampled.txt 2>&1; cat $O/diag_sampled.txt | tail -60
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_configs35.py -m gpu -q -s ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; grep -E "anderson first|passed|failed|FAILED" $O/pytest.log | tail
