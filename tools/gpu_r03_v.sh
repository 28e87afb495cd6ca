#!/bin/bash
# Soak of config 5 with the Anderson step (both regimes), and of configs 2 / 3 on the final tree.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03v; mkdir -p $O; cd /tmp
soak() {
  timeout 900 python $R/bench.py --config $1 --steps $2 --warmup 100 --no-cpu $3 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('soak config $1 $3: $2 control steps timed as one region: %.2f M env-steps/s; state at end %s; regime %s' % (b['value']/1e6, b['state_at_end'], b['config']['regime']))" | tee -a $O/soak.txt
}
soak 5 10000 ""
soak 5 10000 "--atlas-regime collapsing"
soak 2 20000 ""
soak 3 20000 ""
