#!/bin/bash
# tools/exp/ab_flags.sh <tag> <reps> "<script args>" "<flags A>" "<flags B>" ... : like ab_defs.sh, for COMPILER FLAGS of the specialised code object ($RSB_SPEC_EXTRA_FLAGS); "-" = none
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; REPS=$2; SARGS=$3; shift 3
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
: > $O/ab.txt
for r in $(seq $REPS); do
  for d in "$@"; do
    x="$d"; [ "$d" = "-" ] && x=""
    v=$(RSB_SPECIALIZE=compile RSB_SPEC_EXTRA_FLAGS="$x" timeout 300 python tools/exp/pcsample_run.py $SARGS 2>&1 | tail -1)
    echo "[$d] $v" | tee -a $O/ab.txt
  done
done
