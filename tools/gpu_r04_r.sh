#!/bin/bash
mkdir -p gpurun_out/r
echo "--- with the gate" > gpurun_out/r/gate.txt
timeout 120 python tools/exp/pipeline.py --config 2 --steps 300 >> gpurun_out/r/gate.txt 2>&1
echo "--- RSB_X_PIPE_NOGATE=1 (experiment: can deadlock)" >> gpurun_out/r/gate.txt
RSB_X_PIPE_NOGATE=1 timeout 120 python tools/exp/pipeline.py --config 2 --steps 300 >> gpurun_out/r/gate.txt 2>&1
echo "rc=$?" >> gpurun_out/r/gate.txt
grep "^---\|pipelining 1\|rc=" gpurun_out/r/gate.txt | cut -c1-150
