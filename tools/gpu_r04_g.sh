#!/bin/bash
mkdir -p gpurun_out/g
python tools/exp/shards.py --config 2 > gpurun_out/g/shards_c2.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/exp/shards.py --config 2 --shards 4 8 16 > gpurun_out/g/shards_c2_q8.txt 2>&1
python tools/exp/shards.py --config 5 --shards 1 4 8 > gpurun_out/g/shards_c5.txt 2>&1
tail -n 20 gpurun_out/g/*.txt
