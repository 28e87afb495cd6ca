#!/bin/bash
# LDS conflict counters of the resident launch (tools/exp/pcsample_run.py): bash tools/exp/pmc_lds.sh <tag> [script args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc_lds}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o run -- python $R/tools/exp/pcsample_run.py "$@" > $O/p$i.log 2>&1
done
python - $O <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rsb_step_kernel" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} last={v[-1]:.6g}  per wave and control step (K=50, 1024 waves) {v[-1] / 51200:.1f}")
PY
rm -rf $O/p*/
