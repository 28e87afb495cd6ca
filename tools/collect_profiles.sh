#!/bin/bash
# Collect the round's profile evidence on a GPU box (run through gpurun):  bash tools/collect_profiles.sh r01
# Writes under gpurun_out/<tag>/; copy the summaries you want judged into profiles/ (tracked).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python $R/bench.py --max-iter 30 --no-cpu 2>/dev/null | tail -1 > $O/bench_maxiter30.json
python $R/bench.py --no-reset --no-cpu 2>/dev/null | tail -1 > $O/bench_noreset.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu > $O/trace_bench.log 2>&1
# PMC passes: counters only, one group per pass (never combined with trace domains)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python $R/bench.py --no-cpu --steps 50 --warmup 50 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python $R/bench.py --no-cpu --steps 50 --warmup 50 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o run -- python $R/bench.py --no-cpu --steps 50 --warmup 50 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o run -- python $R/bench.py --no-cpu --steps 50 --warmup 50 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = glob.glob("$O/%s/*counter_collection.csv" % sub)
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "rsb_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in acc.items():
        print(sub, k, "mean per dispatch", v / n[k], "dispatches", n[k])
PY
