#!/bin/bash
# alignment sweep: the benchmark instance shifted by n x 4 bytes of s_nop at its entry (everything else identical)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03x; mkdir -p $O; cd /tmp
run() {
  if [ $1 = HEAD ]; then L=$R/raisimlib_amd/lib/librsb.so; else L=$R/raisimlib_amd/lib/librsb.$1.so; fi
  RSB_LIB_PATH=$L python $R/tools/bench_oldlib.py --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-10s %.2f M kernel %.4f ms' % ('$1', b['value']/1e6, b['roofline']['kernel_ms_mean']))" | tee -a $O/sweep.txt
}
for v in ${SWEEP:-base pu7 base pu7 base pu7}; do run $v; done
