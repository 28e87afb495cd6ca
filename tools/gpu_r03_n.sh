#!/bin/bash
# GPU suite + smoke of the tree with the sampled-collider loader (the step kernel is unchanged since tools/gpu_r03_final.sh ran).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
