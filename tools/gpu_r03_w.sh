#!/bin/bash
# same-box A/B of the config-2 rate: libraries of earlier commits (through tools/bench_oldlib.py: missing setters are no-ops) against the tree's
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03w; mkdir -p $O; cd /tmp
for i in 1 2 3; do
  for v in ${AB_LIBS:-b200cbf DRSB_X_SMALLARGS HEAD DRSB_X_ARGPAD256}; do
    if [ $v = HEAD ]; then L=$R/raisimlib_amd/lib/librsb.so; else L=$R/raisimlib_amd/lib/librsb.$v.so; fi
    RSB_LIB_PATH=$L python $R/tools/bench_oldlib.py --no-cpu ${AB_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v  %.2f M kernel %.4f ms' % (b['value']/1e6, b['roofline']['kernel_ms_mean']))" | tee -a $O/bisect.txt
  done
done
