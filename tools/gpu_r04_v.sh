#!/bin/bash
# drop-in path SUSTAINED (2000 control steps: ~1 s, ten cgroup periods) on a 16-CPU quota: threads x how long an idle worker waits actively (RSB_FIBER_SPIN_US)
mkdir -p gpurun_out/v
cat /sys/fs/cgroup/cpu.max > gpurun_out/v/sustained.txt 2>/dev/null
for cfg in "16 200" "16 20" "32 200" "32 20" "32 0" "24 200" "16 200" "32 200"; do
  set -- $cfg
  b0=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  RSB_FIBER_SPIN_US=$2 RSB_FIBER_THREADS=$1 timeout 120 python tools/bench_gym.py 4096 2000 $1 2>/dev/null | python -c "
import json,sys
g=json.loads(sys.stdin.read())
print('$1 threads, spin $2 us: %.2f M env-steps/s, %.3f ms per control step over 2000 control steps' % (g['template_path']['env_steps_per_s']/1e6, g['template_path']['ms_per_control_step']))" | tee -a gpurun_out/v/sustained.txt
  b1=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  echo "   cgroup periods throttled during the run (incl. start-up): $((b1-b0))" | tee -a gpurun_out/v/sustained.txt
done
