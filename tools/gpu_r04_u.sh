#!/bin/bash
# soak: 20 000 control steps (80 000 integrate() per env) pipelined against lock-step, bit-identity of state / contacts / obs at the end
mkdir -p gpurun_out/u
timeout 400 python tools/exp/pipeline.py --config 2 3 --steps 20000 --warmup 100 > gpurun_out/u/soak.txt 2>&1
timeout 300 python tools/exp/pipeline.py --config 5 --steps 8000 --warmup 100 >> gpurun_out/u/soak.txt 2>&1
grep "pipelining" gpurun_out/u/soak.txt | cut -c1-260
