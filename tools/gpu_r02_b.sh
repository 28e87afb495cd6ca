#!/bin/bash
# Round-2 GPU call B: parity after the solver restructuring (per-sweep direction refresh, branch-lean sweep), bench, wave diagnostics.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02b}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_fuzz.py -m gpu -q --durations=12 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --no-cpu 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
timeout 300 python $R/bench.py --no-cpu --config 3 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
timeout 400 python $R/bench.py --no-cpu --config 5 --steps 100 --warmup 50 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
timeout 200 python $R/tools/diag_phases.py > $O/diag_phases.txt 2>&1
timeout 200 python $R/tools/diag_waves.py > $O/diag_waves.txt 2>&1
python - <<PY
import json
for n in ("c2","c3","c5"):
    try:
        b=json.load(open("$O/bench_%s.json"%n)); r=b["roofline"]
        print(n, "%.1fM"%(b["value"]/1e6), "ms/step %.4f"%b["ms_per_step"], "kernel %.4f max %.4f"%(r["kernel_ms_mean"], r["kernel_ms_max"]), b["config"]["regime"], b["state_at_end"])
    except Exception as e: print(n, "FAILED", e)
PY
cat $O/diag_phases.txt; tail -12 $O/diag_waves.txt
