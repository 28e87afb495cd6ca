#!/bin/bash
# round 4, call J: bench.py with pipelined control steps (default), --lockstep, the forced one-rank collective, and the bench-contract tests
mkdir -p gpurun_out/j
( time python bench.py --steps 20 --warmup 5 > gpurun_out/j/bench_driverlike.json 2> gpurun_out/j/bench_driverlike.err ) 2> gpurun_out/j/bench_driverlike.time
python bench.py --no-cpu --no-secondary > gpurun_out/j/bench_c2.json 2> gpurun_out/j/bench_c2.err
python bench.py --no-cpu --no-secondary --lockstep > gpurun_out/j/bench_c2_lockstep.json 2> gpurun_out/j/bench_c2_lockstep.err
python bench.py --no-cpu --no-secondary --force-collective > gpurun_out/j/bench_c2_coll.json 2> gpurun_out/j/bench_c2_coll.err
python bench.py --no-cpu --no-secondary --force-collective --lockstep > gpurun_out/j/bench_c2_coll_lockstep.json 2> gpurun_out/j/bench_c2_coll_lockstep.err
python bench.py --no-cpu --no-secondary --force-collective --obs-exchange peer > gpurun_out/j/bench_c2_peer.json 2> gpurun_out/j/bench_c2_peer.err
for f in gpurun_out/j/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = b["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.2f M" % (b["value"] / 1e6), "ms %.4f" % b["ms_per_step"], "lockstep", (b.get("lockstep") or {}).get("value"), "kernel_ms", r.get("kernel_ms_mean"),
          "pipelined", r.get("pipelined"), "gather:", b["config"]["obs_all_gather"][:60], "| secondary", {k: (v.get("value"), v.get("lockstep_value")) for k, v in (b.get("secondary") or {}).items()},
          "template", (b.get("boundary_template_path") or {}).get("env_steps_per_s"))
except Exception as e:
    print(sys.argv[1], "ERROR", e)
PY
done
cat gpurun_out/j/bench_driverlike.time
tail -3 gpurun_out/j/*.err | head -40
python -m pytest tests/test_gpu_bench_contract.py -m gpu -q > gpurun_out/j/pytest_contract.log 2>&1; tail -8 gpurun_out/j/pytest_contract.log
