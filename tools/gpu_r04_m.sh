#!/bin/bash
mkdir -p gpurun_out/m
show() { python - "$1" "$2" <<'PY'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.2f M (lockstep %.2f M), secondary" % (b["value"] / 1e6, b["lockstep"]["value"] / 1e6), {k: ("%.2f M" % (v["value"] / 1e6), "%.2f M" % (v["lockstep_value"] / 1e6)) for k, v in (b.get("secondary") or {}).items()})
PY
}
GPU_MAX_HW_QUEUES=8 python bench.py --steps 100 --warmup 20 --cpu-seconds 2 > gpurun_out/m/q8.json 2> gpurun_out/m/q8.err; show gpurun_out/m/q8.json "GPU_MAX_HW_QUEUES=8:"
GPU_MAX_HW_QUEUES=2 python bench.py --steps 100 --warmup 20 --cpu-seconds 2 > gpurun_out/m/q2.json 2> gpurun_out/m/q2.err; show gpurun_out/m/q2.json "GPU_MAX_HW_QUEUES=2:"
python bench.py --steps 100 --warmup 20 --cpu-seconds 2 > gpurun_out/m/q4.json 2> gpurun_out/m/q4.err; show gpurun_out/m/q4.json "default:"
python bench.py --config 3 --no-secondary --steps 100 --warmup 20 --cpu-seconds 2 > gpurun_out/m/c3.json 2> gpurun_out/m/c3.err; show gpurun_out/m/c3.json "config 3 alone with the CPU leg:"
