#!/bin/bash
mkdir -p gpurun_out/h
timeout 600 python tools/exp/pipeline.py > gpurun_out/h/pipeline.txt 2>&1
tail -n 20 gpurun_out/h/pipeline.txt
