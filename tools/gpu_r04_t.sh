#!/bin/bash
mkdir -p gpurun_out/t
timeout 200 python tools/exp/pipeline.py --config 2 --envs 1024 2048 8192 16384 32768 65536 --steps 100 --warmup 50 > gpurun_out/t/batch.txt 2>&1
grep "pipelining" gpurun_out/t/batch.txt | cut -c1-120
