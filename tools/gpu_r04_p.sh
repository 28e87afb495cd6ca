#!/bin/bash
# round 4, call P: the full GPU suite on the pipelined tree + the default bench line
mkdir -p gpurun_out/p
python -m pytest tests -m gpu -q > gpurun_out/p/pytest.log 2>&1
tail -6 gpurun_out/p/pytest.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/p/bench_driverlike.json 2> gpurun_out/p/bench_driverlike.err ) 2> gpurun_out/p/bench_driverlike.time
python bench.py > gpurun_out/p/bench_default.json 2> gpurun_out/p/bench_default.err
for f in gpurun_out/p/bench_driverlike.json gpurun_out/p/bench_default.json; do python - $f <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.2f M (lockstep %.2f M), secondary" % (b["value"] / 1e6, b["lockstep"]["value"] / 1e6), {k: ("%.2f M" % (v["value"] / 1e6), "%.2f M" % (v["lockstep_value"] / 1e6)) for k, v in b["secondary"].items()},
      "template %.2f M" % (b["boundary_template_path"]["env_steps_per_s"] / 1e6), "cpu %.2f M" % (b["cpu_baseline"]["value"] / 1e6), "kernel_ms", b["roofline"]["kernel_ms_mean"])
PY
done
grep real gpurun_out/p/bench_driverlike.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
