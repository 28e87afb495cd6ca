#!/bin/bash
# Round 4, call A: two-waves-per-SIMD micro-benchmark + the drop-in path with the hand-written fiber switch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 60 tools/ubench/two_waves > $O/two_waves.txt 2>&1; echo "rc=$?" >> $O/two_waves.txt
cat $O/two_waves.txt
timeout 300 python -m pytest tests/test_cpp_facade.py tests/test_gym_module.py -m gpu -q 2>&1 | tail -5
timeout 300 python tools/bench_gym.py 4096 30 16 > $O/bench_gym_fiber.json 2>$O/bench_gym.err; cat $O/bench_gym_fiber.json
