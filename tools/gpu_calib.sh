#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (separate PMC passes, no trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02calib}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/traffic_calib > $O/known.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $R/tools/ubench/traffic_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $R/tools/ubench/traffic_calib > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for sub, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == cn: acc[r["Kernel_Name"].split("(")[0] + "/" + r["Grid_Size"] if "Grid_Size" in r else r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(cn, k, "per dispatch (KB):", [round(x, 1) for x in v])
PY
cat $O/known.txt
