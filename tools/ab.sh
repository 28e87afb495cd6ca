#!/bin/bash
# Same-box A/B of library variants: tools/ab.sh <reps> <lib-or-dir> ...   ("." = the in-tree library, a directory = another checkout's bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=$1; shift
cd /tmp
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ -d "$R/$v" ]; then B="$R/$v/bench.py"; unset RSB_LIB_PATH; else B="$R/bench.py"; export RSB_LIB_PATH="$R/raisimlib_amd/lib/$v"; fi
    python $B --no-cpu ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-60s %.2f M  kernel %.4f ms' % ('$v', b['value']/1e6, b['roofline']['kernel_ms_mean']))"
  done
done
