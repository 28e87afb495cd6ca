#!/bin/bash
mkdir -p gpurun_out/n
python -m pytest tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/n/pytest.log 2>&1
tail -5 gpurun_out/n/pytest.log
echo "--- XCD-affine hand-over (default)" > gpurun_out/n/xcd.txt
timeout 300 python tools/exp/pipeline.py --config 2 3 5 --steps 300 >> gpurun_out/n/xcd.txt 2>&1
echo "--- RSB_PIPE_XCD=0: agent-scope release / acquire" >> gpurun_out/n/xcd.txt
RSB_PIPE_XCD=0 timeout 300 python tools/exp/pipeline.py --config 2 3 5 --steps 300 >> gpurun_out/n/xcd.txt 2>&1
grep "^---\|pipelining 1" gpurun_out/n/xcd.txt | cut -c1-190
