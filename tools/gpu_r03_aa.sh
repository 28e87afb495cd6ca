#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03aa; mkdir -p $O; cd /tmp
for rep in 1 2 3; do
for e in "" "HSA_ENABLE_INTERRUPT=0"; do
  env $e python $R/bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-26s steps 20: %6.2f M  ms/step %.4f kernel %.4f'%('$e', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_mean']))" | tee -a $O/irq.txt
done; done
for e in "" "HSA_ENABLE_INTERRUPT=0"; do
  env $e python $R/bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-26s default : %6.2f M  ms/step %.4f kernel %.4f'%('$e', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_mean']))" | tee -a $O/irq.txt
done
