#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run of a closed-loop bench: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [max run us]
Finds the action stage's persistent launches and prints, for the shortest one (the timed region of `bench.py --closed-loop-only --steps 20`),
every kernel that overlaps it: start / end / duration in us relative to the stage's start, queue, kind - what a run's fixed cost is made of
(stage launch, gates, first and last step) and whether the steps overlap."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return "step" if "rsb_step_kernel" in n else "stage" if "stage_kernel" in n else "gate" if "gate" in n else n[:36]


pers = [r for r in rows if "stage_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 400e3]
print(len(pers), "persistent stage launches; durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in pers])
if pers:
    p = min(pers, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    s, e = int(p["Start_Timestamp"]), int(p["End_Timestamp"])
    for r in rows:
        if int(r["End_Timestamp"]) >= s - 60e3 and int(r["Start_Timestamp"]) <= e + 60e3:
            print(f"{(int(r['Start_Timestamp']) - s) / 1e3:9.1f} {(int(r['End_Timestamp']) - s) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} q{r.get('Queue_Id', '?')} {short(r['Kernel_Name'])}")
