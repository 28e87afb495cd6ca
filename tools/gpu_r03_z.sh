#!/bin/bash
# alignment sweep of the Atlas-like instance's sweep loop (config 5, standing)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03z; mkdir -p $O; cd /tmp
run() {
  if [ $1 = HEAD ]; then L=$R/raisimlib_amd/lib/librsb.so; else L=$R/raisimlib_amd/lib/librsb.$1.so; fi
  RSB_LIB_PATH=$L python $R/bench.py --no-cpu --config 5 --steps 200 --warmup 50 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-10s %.2f M kernel %.4f ms' % ('$1', b['value']/1e6, b['roofline']['kernel_ms_mean']))" | tee -a $O/sweep.txt
}
for v in ${SWEEP:-HEAD pt0 pt1 pt2 pt3 pt4 pt5 pt6 pt7 HEAD}; do run $v; done
