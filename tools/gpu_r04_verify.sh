#!/bin/bash
# final-tree verification: what the driver runs at round end (GPU suite, smoke, the default bench line at its flags)
mkdir -p gpurun_out/verify
( time python -m pytest tests -m gpu -q ) > gpurun_out/verify/pytest.log 2>&1; tail -5 gpurun_out/verify/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/verify/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/verify/bench.json 2> gpurun_out/verify/bench.err ) 2> gpurun_out/verify/bench.time
python - <<'PY'
import json
b = json.loads(open("gpurun_out/verify/bench.json").read().strip().splitlines()[-1])
print("value %.2f M (lockstep %.2f M), secondary" % (b["value"] / 1e6, b["lockstep"]["value"] / 1e6), {k: ("%.2f M" % (v["value"] / 1e6), "%.2f M" % (v["lockstep_value"] / 1e6)) for k, v in b["secondary"].items()},
      "template %.2f M" % (b["boundary_template_path"]["env_steps_per_s"] / 1e6), b["boundary_template_path"]["attempts_env_steps_per_s"], "cpu %.2f M" % (b["cpu_baseline"]["value"] / 1e6), "hash", b["build"]["source_hash"])
PY
grep real gpurun_out/verify/bench.time
