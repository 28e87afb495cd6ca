#!/bin/bash
# round 4, call I: pipelining - the GPU tests, the price of the agent-scope fences (INCORRECT variants, measurement only), hardware queues
mkdir -p gpurun_out/i
python -m pytest tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/i/pytest.log 2>&1
tail -15 gpurun_out/i/pytest.log
R=$(pwd)
echo "--- default library" > gpurun_out/i/fences.txt
timeout 300 python tools/exp/pipeline.py --config 2 3 5 --steps 300 >> gpurun_out/i/fences.txt 2>&1
for v in 1 2; do
  echo "--- RSB_X_PIPE_FENCE=$v (measurement only)" >> gpurun_out/i/fences.txt
  RSB_LIB_PATH=$R/raisimlib_amd/lib/librsb.DRSB_X_PIPE_FENCE$v.so timeout 300 python tools/exp/pipeline.py --config 2 5 --steps 300 >> gpurun_out/i/fences.txt 2>&1
done
echo "--- default library, GPU_MAX_HW_QUEUES=8" >> gpurun_out/i/fences.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/exp/pipeline.py --config 2 --steps 300 >> gpurun_out/i/fences.txt 2>&1
echo "--- default library, GPU_MAX_HW_QUEUES=2" >> gpurun_out/i/fences.txt
GPU_MAX_HW_QUEUES=2 timeout 300 python tools/exp/pipeline.py --config 2 --steps 300 >> gpurun_out/i/fences.txt 2>&1
grep -v amdgpu.ids gpurun_out/i/fences.txt | grep "^---\|pipelining 1" | cut -c1-200
