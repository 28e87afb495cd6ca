#!/bin/bash
# tools/exp/build_variant.sh -DFLAG [-DFLAG2 ...] : librsb.<tag>.so for an A/B in which only the benchmark's kernel instances (RSB_VARIANT_ONLY,
# default: resident + plain quadruped classes) are compiled with the flags; every other object is the in-tree build's (hard links).  ~25 s instead of ~2 min.
R=$(cd "$(dirname "$0")/../.." && pwd)
ONLY=${RSB_VARIANT_ONLY:-"16,8,64,4 16,8,0,4"}
tag=$(python3 - "$@" <<'PY'
import sys
print("_".join(f.strip("-").replace("=", "") for f in sys.argv[1:]))
PY
)
cd $R/raisimlib_amd/lib/obj
for o in *.o; do
  case "$o" in *.D*|build_stamp*) continue;; esac
  t="${o%.o}.$tag.o"
  [ -e "$t" ] || ln "$o" "$t"
done
for inst in $ONLY; do IFS=, read a b c d <<< "$inst"; rm -f step_${a}_${b}_${c}_${d}_0.$tag.o; done
cd $R
RSB_BUILD_ONLY="$ONLY" python -m raisimlib_amd.build "$@" 2>&1 | grep -v "^/opt/rocm/bin/hipcc\|^g++" | tail -3
ls -la raisimlib_amd/lib/librsb.$tag.so
