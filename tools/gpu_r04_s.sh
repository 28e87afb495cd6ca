#!/bin/bash
# pipelined control steps under a profiler that serialises dispatches (rocprofv3 --pmc): the library keeps them in lock-step.  Bounded: 80 s.
mkdir -p gpurun_out/s
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_gpu_pipeline.py -m gpu -q > $R/gpurun_out/s/pytest.log 2>&1; tail -3 $R/gpurun_out/s/pytest.log
cd /tmp && export TMPDIR=/tmp
( time timeout 80 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/s/pmc -o run -- python $R/bench.py --no-cpu --no-secondary --steps 10 --warmup 5 --preroll 10 ) > $R/gpurun_out/s/pmc.log 2>&1
echo "rc=$?" >> $R/gpurun_out/s/pmc.log
grep -v "^W2026\|^I2026\|^E2026" $R/gpurun_out/s/pmc.log | tail -8 | cut -c1-400
