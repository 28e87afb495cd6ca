#!/bin/bash
# Round-3 GPU call A: full GPU test-suite of the new tree, then the bench configurations (config 2 default + driver-like,
# config 3 shared / per-env maps, config 5 standing / collapsing).  Writes under gpurun_out/r03a/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $O/bench_c2_driverlike.json
timeout 300 python $R/bench.py --config 3 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
timeout 400 python $R/bench.py --config 3 --per-env-maps --no-cpu 2>$O/bench_c3pe.err | tail -1 > $O/bench_c3_per_env_maps.json
timeout 400 python $R/bench.py --config 5 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
timeout 400 python $R/bench.py --config 5 --atlas-regime collapsing 2>$O/bench_c5c.err | tail -1 > $O/bench_c5_collapsing.json
timeout 200 python $R/bench.py --force-collective --no-cpu 2>/dev/null | tail -1 > $O/bench_c2_force_collective.json
python - <<PY
import json
for n in ("c2","c2_driverlike","c3","c3_per_env_maps","c5","c5_collapsing","c2_force_collective"):
    try:
        b=json.load(open("$O/bench_%s.json"%n)); r=b["roofline"]
        print(n, "%.2fM"%(b["value"]/1e6), "ms/step %.4f"%b["ms_per_step"], "kernel %.4f"%r["kernel_ms_mean"], b["config"]["regime"], b["state_at_end"], b.get("cpu_baseline",{}).get("value"), b.get("cpu_baseline",{}).get("cores"))
    except Exception as e: print(n, "FAILED", e)
PY
ls $O
