#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -m gpu -q -k "valley or two_contacts or heightmap or ridge" ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -40 $O/pytest.log | cut -c1-250
