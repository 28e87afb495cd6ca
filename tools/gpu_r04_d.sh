#!/bin/bash
# Round 4, call D: what a 256-register cap (two waves per SIMD by registers, the rest spilled to scratch) does to the EXISTING kernel at 32 lanes per env,
# at the benchmark's batch and at larger ones (the why-not file's "larger batches" claim, measured).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # label, lib, extra args
  RSB_LIB_PATH=$2 timeout 300 python $R/bench.py --no-cpu --steps 100 --warmup 30 ${@:3} 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-64s %8.2f M env-steps/s  kernel %.4f ms' % ('$1', b['value']/1e6, b['roofline']['kernel_ms_mean']))" | tee -a $O/wpe2.txt
}
D=$R/raisimlib_amd/lib/librsb.so
W=$R/raisimlib_amd/lib/librsb.DRSB_X_WPE2.so
for n in 4096 8192 16384 32768; do
  run "shipped kernel, 16 lanes per env (415 regs), N = $n" $D --envs-per-gpu $n
  run "shipped kernel, 32 lanes per env (one wave per SIMD), N = $n" $D --envs-per-gpu $n --lanes-per-env 32
  run "256-register cap, 32 lanes per env (two waves per SIMD), N = $n" $W --envs-per-gpu $n --lanes-per-env 32
done
