#!/bin/bash
# Round-3 GPU call F: peer-mapped obs exchange with the flags written by the stream (no fence in the kernel): tests, then config 2
# with no collective / RCCL all-gather / peer exchange forced onto one rank.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
for t in one_rank two_worlds two_processes; do
  ( timeout 200 python -m pytest tests/test_gpu_obs_peer.py -m gpu -q -x -k $t ) > $O/pytest_peer_$t.log 2>&1
  echo "peer test $t rc=$?"; tail -3 $O/pytest_peer_$t.log
done
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  for v in "" "--force-collective" "--force-collective --obs-exchange peer"; do
    timeout 200 python $R/bench.py --no-cpu --steps 200 --warmup 50 $v 2>$O/bench.err | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-45s %.2f M  ms/step %.4f kernel %.4f ms  | %s' % ('$v', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_ms_mean'], b['config']['obs_all_gather'][:40]))" 2>&1 | tee -a $O/collective_ab.txt
  done
done
tail -3 $O/bench.err
