#!/bin/bash
# round 4, call E: exact box x height map on the device (KATs + parity), and the benchmark instance beside it
mkdir -p gpurun_out/e
python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -m gpu -q -s -k "box or capsule or cylinder or sampled or second" > gpurun_out/e/pytest.log 2>&1
tail -40 gpurun_out/e/pytest.log
python bench.py --no-cpu --no-secondary --steps 100 --warmup 30 > gpurun_out/e/bench_c2.json 2> gpurun_out/e/bench_c2.err
cat gpurun_out/e/bench_c2.json | cut -c1-400
