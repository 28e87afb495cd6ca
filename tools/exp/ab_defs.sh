#!/bin/bash
# tools/exp/ab_defs.sh <tag> <reps> "<script args>" "<defs A>" "<defs B>" ... : same-box A/B of kernel variants through the specialisation path - every variant is the
# workload of tools/exp/pcsample_run.py with RSB_SPEC_EXTRA_DEFS=<defs> (compiled on the box, ~3 s each, cached); "-" = no extra flags.  One value per line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; REPS=$2; SARGS=$3; shift 3
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
: > $O/ab.txt
for r in $(seq $REPS); do
  for d in "$@"; do
    x="$d"; [ "$d" = "-" ] && x=""
    v=$(RSB_SPECIALIZE=compile RSB_SPEC_EXTRA_DEFS="$x" timeout 300 python tools/exp/pcsample_run.py $SARGS 2>&1 | tail -1)
    echo "[$d] $v" | tee -a $O/ab.txt
  done
done
