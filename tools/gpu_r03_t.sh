#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
for v in DRSB_X_HMSTAMP1 DRSB_X_HMSTAMP2; do
  RSB_LIB_PATH=$R/raisimlib_amd/lib/librsb.$v.so timeout 600 python tools/diag_config3_phases.py 2>&1 | grep -A14 "config 3" | grep -E "collision: terrain|stamp 15" | sed "s/^/$v /" | tee -a $O/hm_steps.txt
done
