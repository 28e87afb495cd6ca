#!/bin/bash
# The round's evidence call (one gpurun lease, ~12 min): GPU test-suite, smoke, the default bench line at the driver's flags and at the default flags, configs 3 / 5,
# the closed loop, rocprofv3 kernel traces, PMC passes (one counter group per pass, never combined with trace domains).
# tools/summarise_r06.py <tag> turns gpurun_out/<tag>/ into the tracked profiles/<tag>_* files.
TAG=${1:-r06}
B="--no-cpu --no-secondary"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
STEP_TIMEOUT=${STEP_TIMEOUT:-900} bash tools/lease.sh $TAG \
  "pytest@pytest:tests -m gpu" "smoke@smoke" \
  "bench@default20:--steps 20 --warmup 5" "bench@default300:" \
  "bench@c3:--config 3" "bench@c5:--config 5" \
  "bench@closed20:--closed-loop-only --steps 20 --warmup 5" "bench@closed300:--closed-loop-only" \
  "trace@trace_c2:$B --steps 20 --warmup 5" "trace@trace_c3:$B --config 3 --steps 20 --warmup 5" "trace@trace_c5:$B --config 5 --steps 20 --warmup 5" "trace@trace_closed:--closed-loop-only --steps 20 --warmup 5" \
  "pmc@pmc_fetch_c2:FETCH_SIZE:$B --steps 50 --warmup 10" "pmc@pmc_write_c2:WRITE_SIZE:$B --steps 50 --warmup 10" \
  "pmc@pmc_fetch_c3:FETCH_SIZE:$B --config 3 --steps 50 --warmup 10" "pmc@pmc_write_c3:WRITE_SIZE:$B --config 3 --steps 50 --warmup 10" \
  "pmc@pmc_fetch_c5:FETCH_SIZE:$B --config 5 --steps 30 --warmup 10" "pmc@pmc_write_c5:WRITE_SIZE:$B --config 5 --steps 30 --warmup 10" \
  "pmc@pmc_sq:$SQ1:$B --steps 50 --warmup 10" "pmc@pmc_sq2:$SQ2:$B --steps 50 --warmup 10" \
  "py@prof_template:tools/prof_template_path.py 4096 400 16"
