#!/bin/bash
mkdir -p gpurun_out/q
R=$(pwd)
echo "--- default (contacts 0-4 straight in the exchange)" > gpurun_out/q/straight.txt
timeout 300 python tools/exp/pipeline.py --config 2 3 --steps 300 >> gpurun_out/q/straight.txt 2>&1
echo "--- RSB_X_STRAIGHT_PIPE=4 (pipelined classes: contacts 0-3 straight, the fifth behind a test)" >> gpurun_out/q/straight.txt
RSB_LIB_PATH=$R/raisimlib_amd/lib/librsb.DRSB_X_STRAIGHT_PIPE4.so timeout 300 python tools/exp/pipeline.py --config 2 3 --steps 300 >> gpurun_out/q/straight.txt 2>&1
grep "^---\|pipelining 1" gpurun_out/q/straight.txt | cut -c1-150
