#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
timeout 600 python tools/diag_config3_phases.py > $O/diag_c3.txt 2>&1; cat $O/diag_c3.txt
