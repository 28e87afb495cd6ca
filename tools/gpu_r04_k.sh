#!/bin/bash
mkdir -p gpurun_out/k
for cfg in 2 3; do
for st in 20 50 100 300; do
  for extra in "" "--no-kernel-events"; do
    python bench.py --no-cpu --no-secondary --config $cfg --steps $st --warmup 5 $extra 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('config $cfg steps $st $extra: value %.2f M, ms/step %.4f, host enqueue %.4f, lockstep %.2f M' % (b['value']/1e6, b['ms_per_step'], b['host_enqueue_ms_per_step'], b['lockstep']['value']/1e6))" | tee -a gpurun_out/k/steps_sweep.txt
  done
done
done
