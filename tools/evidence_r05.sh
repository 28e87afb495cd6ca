#!/bin/bash
# The round's evidence call (one gpurun lease, ~10 min): GPU test-suite, smoke, the default bench line at the driver's flags and at the default flags,
# configs 2 / 3 / 5, the closed loop, rocprofv3 kernel traces, PMC passes (one counter group per pass, never combined with trace domains).
# tools/summarise_r05.py <tag> turns gpurun_out/<tag>/ into the tracked profiles/<tag>_* files.
TAG=${1:-r05}
B="--no-cpu --no-secondary"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
STEP_TIMEOUT=${STEP_TIMEOUT:-900} bash tools/lease.sh $TAG \
  "pytest@pytest:tests -m gpu" "smoke@smoke" \
  "bench@default20:--steps 20 --warmup 5" "bench@default300:" \
  "bench@c2:$B" "bench@c3:$B --config 3" "bench@c5:$B --config 5" "bench@c2_lockstep:$B --lockstep" \
  "bench@closed20:--closed-loop-only --steps 20 --warmup 5" "bench@closed300:--closed-loop-only" \
  "trace@trace_c2:$B" "trace@trace_c3:$B --config 3" "trace@trace_c5:$B --config 5" "trace@trace_closed:--closed-loop-only" \
  "pmc@pmc_fetch_c2:FETCH_SIZE:$B --steps 50 --warmup 50" "pmc@pmc_write_c2:WRITE_SIZE:$B --steps 50 --warmup 50" \
  "pmc@pmc_fetch_c3:FETCH_SIZE:$B --config 3 --steps 50 --warmup 50" "pmc@pmc_write_c3:WRITE_SIZE:$B --config 3 --steps 50 --warmup 50" \
  "pmc@pmc_fetch_c5:FETCH_SIZE:$B --config 5 --steps 30 --warmup 30" "pmc@pmc_write_c5:WRITE_SIZE:$B --config 5 --steps 30 --warmup 30" \
  "pmc@pmc_sq:$SQ1:$B --lockstep --steps 50 --warmup 50" "pmc@pmc_sq2:$SQ2:$B --lockstep --steps 50 --warmup 50" \
  "pmc@pmc_sq_c5:SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY:$B --lockstep --config 5 --steps 30 --warmup 30"
