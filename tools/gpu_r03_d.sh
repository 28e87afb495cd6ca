#!/bin/bash
# Round-3 GPU call D: the exchange of a pass restricted to the slots that changed (default build) against the exchange of every
# slot (librsb.DRSB_X_NOMASK.so), same box: config 2, config 5 standing / collapsing; parity tests on the default build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q -k "parity or population or fuzz or golden or kat" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
AB_ARGS="--steps 200 --warmup 50" bash $R/tools/ab.sh 3 librsb.so librsb.DRSB_X_NOMASK.so > $O/ab_mask_c2.txt 2>&1
cat $O/ab_mask_c2.txt
AB_ARGS="--config 5 --steps 100 --warmup 20" bash $R/tools/ab.sh 2 librsb.so librsb.DRSB_X_NOMASK.so > $O/ab_mask_c5.txt 2>&1
cat $O/ab_mask_c5.txt
AB_ARGS="--config 5 --atlas-regime collapsing --steps 100 --warmup 20" bash $R/tools/ab.sh 2 librsb.so librsb.DRSB_X_NOMASK.so > $O/ab_mask_c5c.txt 2>&1
cat $O/ab_mask_c5c.txt
AB_ARGS="--config 3 --steps 200 --warmup 50" bash $R/tools/ab.sh 2 librsb.so librsb.DRSB_X_NOMASK.so > $O/ab_mask_c3.txt 2>&1
cat $O/ab_mask_c3.txt
