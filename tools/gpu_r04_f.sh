#!/bin/bash
# round 4, call F: per-env parity at N = 4096 on the config-3 and config-5 populations
mkdir -p gpurun_out/f
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "populations_at_4096" > gpurun_out/f/pytest.log 2>&1
tail -60 gpurun_out/f/pytest.log
