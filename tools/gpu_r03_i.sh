#!/bin/bash
# Round-3 GPU call I: peer obs exchange published by the step kernel's last wave (write-through stores, relaxed atomics, no
# fence): tests, then the one-rank cost against no exchange / RCCL, with and without the consumer's wait.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
for t in one_rank two_worlds two_processes; do
  ( timeout 200 python -m pytest tests/test_gpu_obs_peer.py -m gpu -q -x -k $t ) > $O/pytest_peer_$t.log 2>&1
  echo "peer test $t rc=$?"; tail -3 $O/pytest_peer_$t.log
done
cd /tmp && export TMPDIR=/tmp
run() { label=$1; shift; env "$@" 2>$O/bench.err | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-60s %.2f M  ms/step %.4f kernel %.4f ms' % ('$label', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_ms_mean']))" 2>&1 | tee -a $O/peer_ab.txt; }
B="python $R/bench.py --no-cpu --steps 200 --warmup 50"
for i in 1 2 3; do
  run "no exchange" A=1 $B
  run "RCCL all-gather, in line" A=1 $B --force-collective
  run "peer: in-kernel publication + wait packet" A=1 $B --force-collective --obs-exchange peer
  run "peer: in-kernel publication + wait kernel" RSB_OBS_PEER_WAIT_KERNEL=1 $B --force-collective --obs-exchange peer
  run "peer: in-kernel publication, no wait" A=1 $B --force-collective --obs-exchange peer --peer-no-wait
done
