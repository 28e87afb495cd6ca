#!/bin/bash
# Round-3 GPU call M: the final tree once more - whole GPU suite, smoke, the driver's bench command - and soak runs (bench.py with a long
# timed region: the populations stay stationary, no non-finite state).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driverlike.json
python -c "import json; b=json.load(open('$O/bench_driverlike.json')); print('driver-like: %.2f M, ms/step %.4f, kernel %.4f, cpu %.2f M on %d threads'%(b['value']/1e6,b['ms_per_step'],b['roofline']['kernel_ms_mean'],b['cpu_baseline']['value']/1e6,b['cpu_baseline']['cores']))"
for cfg in "2 20000" "3 20000" "5 5000"; do
  set -- $cfg
  timeout 600 python $R/bench.py --config $1 --steps $2 --warmup 100 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('soak config $1: $2 control steps timed as one region: %.2f M env-steps/s; state at end %s; regime %s' % (b['value']/1e6, b['state_at_end'], b['config']['regime']))" | tee -a $O/soak.txt
done
