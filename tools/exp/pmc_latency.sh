#!/bin/bash
# Counter passes over the resident launch (tools/exp/pcsample_run.py): average latency of scalar loads / LDS ops of a lone wave, instruction-fetch stalls.
# usage (GPU box): bash tools/exp/pmc_latency.sh <tag> [script args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc_lat}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o run -- python $R/tools/exp/pcsample_run.py "$@" > $O/p$i.log 2>&1
  echo "rc=$? $grp" >> $O/p$i.log
done
python - $O <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/summary.txt", "w") as out:
    for k, d in agg.items():
        out.write(k + "\n")
        for c, v in sorted(d.items()):
            out.write(f"  {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g} last={v[-1]:.6g}\n")
print(open(O + "/summary.txt").read())
PY
rm -rf $O/p*/   # raw csv not needed
