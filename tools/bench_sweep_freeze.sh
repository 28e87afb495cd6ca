#!/bin/bash
# Diagnostic: config-2 bench at several freeze_after values (sweeps before friction directions lag).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for f in "$@"; do
  python $R/bench.py --no-cpu --freeze-after $f 2>/dev/null | tail -1 > /tmp/b_$f.json
  python - <<PY
import json
b = json.load(open("/tmp/b_$f.json"))
print("freeze_after", $f, "%.1fM" % (b["value"] / 1e6), "kernel %.4f p50 %.4f max %.4f" % (b["roofline"]["kernel_ms_mean"], b["roofline"]["kernel_ms_p50"], b["roofline"]["kernel_ms_max"]), b["state_at_end"])
PY
done
