#!/bin/bash
# Round-3 GPU call C: whole GPU suite (packed Delassus layout for kmax 16, cheaper policy selection), A/B against the round-2
# head on config 2, config 5 at two envs per wave (LPE 32, the new default) and one (LPE 64).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
AB_ARGS="--steps 200 --warmup 50" bash $R/tools/ab.sh 3 . _ab_head > $O/ab_head.txt 2>&1
cat $O/ab_head.txt
for regime in standing collapsing; do
  for lpe in 0 64; do
    timeout 400 python $R/bench.py --config 5 --atlas-regime $regime --lanes-per-env $lpe --no-cpu 2>$O/c5_${regime}_$lpe.err | tail -1 > $O/c5_${regime}_lpe$lpe.json
    python - <<PY
import json
try:
    b=json.load(open("$O/c5_${regime}_lpe$lpe.json")); print("config 5 $regime lpe $lpe -> %d: %.2f M, kernel %.4f ms"%(b["config"]["lanes_per_env"], b["value"]/1e6, b["roofline"]["kernel_ms_mean"]), b["state_at_end"])
except Exception as e: print("config 5 $regime lpe $lpe FAILED", e); print(open("$O/c5_${regime}_$lpe.err").read()[-600:])
PY
  done
done
