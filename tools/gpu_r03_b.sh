#!/bin/bash
# Round-3 GPU call B: re-run of the GPU tests that failed in call A + the gym module tests, same-box A/B of the current kernel
# against the round-2 head (config 2), phase / wave diagnostics with the finer stamps, throughput of the Python gym boundary.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q -k "comm_launcher or atlas_parity or self_collision_parity or uninitialised or gym or control_step_equals or api_errors or world_x" ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
AB_ARGS="--steps 200 --warmup 50" bash $R/tools/ab.sh 3 . _ab_head > $O/ab_head.txt 2>&1
cat $O/ab_head.txt
timeout 300 python $R/tools/diag_phases.py > $O/diag_phases.txt 2>&1
cat $O/diag_phases.txt
RSB_PROF_FINE=1 timeout 300 python $R/tools/diag_waves.py > $O/diag_waves_fine.txt 2>&1
tail -12 $O/diag_waves_fine.txt
timeout 300 python $R/tools/bench_gym.py 4096 30 > $O/bench_gym.json 2>$O/bench_gym.err
cat $O/bench_gym.json; tail -3 $O/bench_gym.err
