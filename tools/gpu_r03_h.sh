#!/bin/bash
# Round-3 GPU call H: what the peer obs exchange's 9 us per control step are made of (stream memory-write / memory-wait packets
# against one-wave kernels, producer side alone), and the MFMA Delassus micro-benchmark.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # label, env assignments..., -- bench args
  label=$1; shift
  env "$@" 2>$O/bench.err | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-60s %.2f M  ms/step %.4f kernel+packets %.4f ms' % ('$label', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_ms_mean']))" 2>&1 | tee -a $O/peer_breakdown.txt
}
B="python $R/bench.py --no-cpu --steps 200 --warmup 50"
for i in 1 2; do
  run "no exchange" A=1 $B
  run "peer: write packet + wait packet" A=1 $B --force-collective --obs-exchange peer
  run "peer: write packet, no wait" A=1 $B --force-collective --obs-exchange peer --peer-no-wait
  run "peer: flag kernel, no wait" RSB_OBS_PEER_FLAG_KERNEL=1 $B --force-collective --obs-exchange peer --peer-no-wait
  run "peer: flag kernel + wait packet" RSB_OBS_PEER_FLAG_KERNEL=1 $B --force-collective --obs-exchange peer
  run "peer: flag kernel + wait kernel" RSB_OBS_PEER_FLAG_KERNEL=1 RSB_OBS_PEER_WAIT_KERNEL=1 $B --force-collective --obs-exchange peer
  run "peer: write packet + wait kernel" RSB_OBS_PEER_WAIT_KERNEL=1 $B --force-collective --obs-exchange peer
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfd $R/tools/ubench/mfma_delassus.hip 2>/dev/null && /tmp/mfd | tee $O/ubench_mfma_delassus.txt
