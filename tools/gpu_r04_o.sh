#!/bin/bash
mkdir -p gpurun_out/o
R=$(pwd)
echo "--- default (tables first, then the wait)" > gpurun_out/o/order.txt
timeout 300 python tools/exp/pipeline.py --config 2 3 --steps 300 >> gpurun_out/o/order.txt 2>&1
echo "--- RSB_X_PIPE_ORDER=1 (wait and state loads first, then the tables)" >> gpurun_out/o/order.txt
RSB_LIB_PATH=$R/raisimlib_amd/lib/librsb.DRSB_X_PIPE_ORDER1.so timeout 300 python tools/exp/pipeline.py --config 2 3 --steps 300 >> gpurun_out/o/order.txt 2>&1
echo "--- RSB_X_PIPE_ORDER=1, RSB_PIPE_XCD=0" >> gpurun_out/o/order.txt
RSB_PIPE_XCD=0 RSB_LIB_PATH=$R/raisimlib_amd/lib/librsb.DRSB_X_PIPE_ORDER1.so timeout 300 python tools/exp/pipeline.py --config 2 --steps 300 >> gpurun_out/o/order.txt 2>&1
grep "^---\|pipelining" gpurun_out/o/order.txt | cut -c1-150
