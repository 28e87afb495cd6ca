#!/bin/bash
# Round-2 GPU call A (through gpurun): full GPU test-suite, then the three bench configurations and kernel traces.
# Writes under gpurun_out/r02a/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
timeout 300 python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c2_driverlike.json
timeout 300 python $R/bench.py --config 3 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
timeout 400 python $R/bench.py --config 5 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
timeout 300 python $R/bench.py --no-cpu --settle-tol 1e-4 2>/dev/null | tail -1 > $O/bench_c2_settle1e-4.json
for c in 2 5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c$c -o bench -- python $R/bench.py --no-cpu --config $c > $O/trace_c$c.log 2>&1
done
timeout 200 python $R/tools/diag_phases.py > $O/diag_phases.txt 2>&1
timeout 200 python $R/tools/diag_waves.py > $O/diag_waves.txt 2>&1
python - <<PY
import json
for n in ("c2","c2_driverlike","c3","c5","c2_settle1e-4"):
    try:
        b=json.load(open("$O/bench_%s.json"%n)); r=b["roofline"]
        print(n, "%.1fM"%(b["value"]/1e6), "ms/step %.4f"%b["ms_per_step"], "kernel %.4f raw %.4f ovh %.4f in-region %s"%(r["kernel_ms_mean"], r["kernel_ms_mean_bracket_raw"], r["event_pair_overhead_ms"], r["timed_region_brackets"]), b["config"]["regime"], b["state_at_end"], b.get("cpu_baseline",{}).get("value"), b.get("cpu_baseline",{}).get("cores"))
    except Exception as e: print(n, "FAILED", e)
PY
ls $O
