#!/bin/bash
# Round-3 GPU call G: threaded fibers (facade + gym tests, gym throughput at 1 / 8 / 16 / 32 host threads), peer obs exchange with a
# coarse-grained gathered buffer (diagnostic) against the fine-grained one and RCCL.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_cpp_facade.py tests/test_gym_module.py tests/test_gpu_obs_peer.py -m gpu -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
nproc
cd /tmp && export TMPDIR=/tmp
for t in 1 8 16 32; do
  timeout 300 python $R/tools/bench_gym.py 4096 30 $t 2>$O/gym_$t.err | tail -1 > $O/bench_gym_t$t.json
  python -c "import json; b=json.load(open('$O/bench_gym_t$t.json')); print('gym threads $t: template %.2f M env-steps/s (%.2f ms / control step), device env host buffers %.1f M'%(b['template_path']['env_steps_per_s']/1e6, b['template_path']['ms_per_control_step'], b['device_env_host_buffers']['env_steps_per_s']/1e6))"
done
for i in 1 2; do
  for v in "" "--force-collective" "--force-collective --obs-exchange peer"; do
    timeout 200 python $R/bench.py --no-cpu --steps 200 --warmup 50 $v 2>$O/bench.err | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-45s %.2f M  ms/step %.4f kernel %.4f ms' % ('fine   $v', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_ms_mean']))" 2>&1 | tee -a $O/collective_ab.txt
  done
  RSB_OBS_PEER_COARSE=1 timeout 200 python $R/bench.py --no-cpu --steps 200 --warmup 50 --force-collective --obs-exchange peer 2>$O/bench.err | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%-45s %.2f M  ms/step %.4f kernel %.4f ms' % ('COARSE --force-collective --obs-exchange peer', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_ms_mean']))" 2>&1 | tee -a $O/collective_ab.txt
done
