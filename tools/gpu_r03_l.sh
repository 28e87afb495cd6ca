#!/bin/bash
# Round-3 GPU call L: LDS layout variants (slot pitches / per-env pad: the bank pattern changes, the instruction stream does not),
# config 2, same box; the gym path with the persistent thread pool.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V="librsb.so librsb.DRSB_X_UPSLOT32.so librsb.DRSB_X_UPSLOT36.so librsb.DRSB_X_BODYSLOT28.so librsb.DRSB_X_ENVPAD4.so librsb.DRSB_X_ENVPAD12.so librsb.DRSB_X_UPSLOT32_DRSB_X_ENVPAD12.so"
AB_ARGS="--steps 200 --warmup 50" bash $R/tools/ab.sh 2 $V > $O/ab_layout.txt 2>&1
cat $O/ab_layout.txt
for v in librsb.so librsb.DRSB_X_UPSLOT32.so librsb.DRSB_X_BODYSLOT28.so librsb.DRSB_X_ENVPAD12.so; do
  RSB_LIB_PATH=$R/raisimlib_amd/lib/$v timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_$v -o run -- python $R/bench.py --no-cpu --steps 50 --warmup 50 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_$v/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if "rsb_step_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
m = {k: acc[k] / n[k] for k in acc}
print("%-50s bank conflicts %.2f %% of wave cycles, s_waitcnt %.1f %%" % ("$v", 100 * m["SQ_LDS_BANK_CONFLICT"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
done 2>&1 | tee $O/pmc_layout.txt
cd $R
( timeout 600 python -m pytest tests/test_cpp_facade.py tests/test_gym_module.py -m gpu -q ) 2>&1 | tail -3
cd /tmp
timeout 300 python $R/tools/bench_gym.py 4096 40 16 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('gym, persistent pool, 16 threads: template %.2f M env-steps/s (%.2f ms per control step)' % (b['template_path']['env_steps_per_s']/1e6, b['template_path']['ms_per_control_step']))"
