#!/bin/bash
# Round 4, call C: the new GPU tests (capsule contacts, per-env parity at N = 4096) first, then the whole GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -m gpu -q -x -s -k "capsule or benchmark_population_at_4096" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log; tail -25 $O/pytest_new.log
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do timeout 300 python $R/bench.py --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('c2 %.2fM kernel %.4f' % (b['value']/1e6, b['roofline']['kernel_ms_mean']))"; done
