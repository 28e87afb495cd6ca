#!/bin/bash
# GPU test-suite + smoke of the tree (and, when raisimlib_amd/lib/librsb.base.so exists, a same-box A/B of config 2 against it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
if [ -f raisimlib_amd/lib/librsb.base.so ]; then SWEEP="base HEAD base HEAD base HEAD" bash tools/gpu_r03_x.sh; fi
