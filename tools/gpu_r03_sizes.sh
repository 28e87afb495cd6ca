#!/bin/bash
# config-2 workload at other batch sizes (not the headline: the configuration is 4096 envs per GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03sizes; mkdir -p $O; cd /tmp
for n in 1024 2048 4096 8192 16384 32768 65536; do
  timeout 300 python $R/bench.py --no-cpu --envs-per-gpu $n --steps 150 --warmup 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('envs %6d: %7.2f M env-steps/s, %.4f ms per control step, kernel %.4f ms'%($n, d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_mean']))" | tee -a $O/sizes.txt
done
for n in 2048 4096 8192 16384; do
  timeout 300 python $R/bench.py --no-cpu --config 5 --envs-per-gpu $n --steps 100 --warmup 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config 5 envs %6d: %7.2f M env-steps/s, kernel %.4f ms'%($n, d['value']/1e6, d['roofline']['kernel_ms_mean']))" | tee -a $O/sizes.txt
done
