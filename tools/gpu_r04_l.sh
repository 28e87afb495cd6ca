#!/bin/bash
mkdir -p gpurun_out/l
for i in 1 2; do
( time python bench.py --steps 20 --warmup 5 > gpurun_out/l/bench_driverlike_$i.json 2> gpurun_out/l/bench_driverlike_$i.err ) 2> gpurun_out/l/bench_driverlike_$i.time
python - gpurun_out/l/bench_driverlike_$i.json <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-like: value %.2f M (lockstep %.2f M), secondary" % (b["value"] / 1e6, b["lockstep"]["value"] / 1e6), {k: ("%.2f M" % (v["value"] / 1e6), "%.2f M" % (v["lockstep_value"] / 1e6)) for k, v in b["secondary"].items()},
      "template %.2f M" % (b["boundary_template_path"]["env_steps_per_s"] / 1e6), "cpu %.2f M" % (b["cpu_baseline"]["value"] / 1e6))
PY
cat gpurun_out/l/bench_driverlike_$i.time | grep real
done
python bench.py > gpurun_out/l/bench_default.json 2> gpurun_out/l/bench_default.err
python - gpurun_out/l/bench_default.json <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("default: value %.2f M (lockstep %.2f M), secondary" % (b["value"] / 1e6, b["lockstep"]["value"] / 1e6), {k: ("%.2f M" % (v["value"] / 1e6), "%.2f M" % (v["lockstep_value"] / 1e6)) for k, v in b["secondary"].items()},
      "template %.2f M" % (b["boundary_template_path"]["env_steps_per_s"] / 1e6), "cpu %.2f M" % (b["cpu_baseline"]["value"] / 1e6))
PY
