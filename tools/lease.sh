#!/bin/bash
# tools/lease.sh <tag> <step> [<step> ...] - ONE parameterised script for a GPU call (replaces rounds 1-4's ~60 one-off tools/gpu_r0N_x.sh files).
# Run through gpurun from the repo root:   gpurun --timeout 900 -- 'bash tools/lease.sh r05_a "pytest:tests/test_gpu_closed_loop.py" "bench:--steps 20 --warmup 5"'
# Every step has a name:argument form (name@label:argument names its files NN_<label>.* instead of NN_<name>.*), runs under its own `timeout`, and logs to gpurun_out/<tag>/NN_<name>.{log,json} (merged back by gpurun).
#   pytest:<pytest args>        python -m pytest <args> -q -x --timeout 600             (pytest:-m gpu tests = the GPU tier)
#   smoke                       __graft_entry__.smoke()
#   bench:<bench.py args>       python bench.py <args>: last stdout line -> .json, stderr -> .log
#   trace:<bench.py args>       rocprofv3 --kernel-trace --stats over bench.py <args> (csv under NN_trace/)
#   pmc:<COUNTERS>:<bench args> rocprofv3 --pmc <COUNTERS (space separated)> over bench.py <args> - counters only, never with trace domains
#   py:<script and args>        python <script and args>   (tools/diag_*.py, tools/exp/*.py)
#   sh:<command>                bash -c <command>
#   ab:<reps>:<lib>,<lib>:<bench args>   same-box A/B: librsb variants (RSB_LIB_PATH) alternating, `reps` rounds, one value per line
# STEP_TIMEOUT (seconds, default 600) bounds every step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
T=${STEP_TIMEOUT:-600}
i=0
for step in "$@"; do
  i=$((i + 1)); n=$(printf %02d $i)
  kind=${step%%:*}; arg=${step#*:}; [ "$kind" = "$step" ] && arg=""
  label=""; case "$kind" in *@*) label=${kind#*@}; kind=${kind%%@*};; esac      # kind@label:arg names the step's files NN_<label>.* (tools/summarise_r05.py finds them by label)
  t0=$(date +%s)
  case "$kind" in
    pytest) ( timeout $T python -m pytest $arg -q -x --timeout 600 --durations=6 ) > "$O/${n}_${label:-pytest}.log" 2>&1; echo "rc=$?" >> "$O/${n}_${label:-pytest}.log"; tail -6 "$O/${n}_${label:-pytest}.log" ;;
    smoke)  timeout $T python -c "import __graft_entry__ as g; g.smoke()" > "$O/${n}_${label:-smoke}.log" 2>&1; echo "rc=$?" >> "$O/${n}_${label:-smoke}.log"; tail -2 "$O/${n}_${label:-smoke}.log" ;;
    bench)  timeout $T python bench.py $arg 2> "$O/${n}_${label:-bench}.log" | tail -1 > "$O/${n}_${label:-bench}.json"; echo "rc=${PIPESTATUS[0]} args: $arg" >> "$O/${n}_${label:-bench}.log"
            python - "$O/${n}_${label:-bench}.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d.get("closed_loop") or {}
    print("bench:", {k: (round(v) if isinstance(v, float) and v > 1e3 else v) for k, v in d.items() if k in ("value", "ms_per_step", "n_gpus")},
          "lockstep", round(((d.get("lockstep") or {}).get("value") or 0)),
          "closed_loop", {m: round((c.get(m) or {}).get("value") or 0) for m in ("pipelined", "lockstep")} if c else None,
          "mlp", {m: round(((c.get("mlp") or {}).get(m) or {}).get("value") or 0) for m in ("pipelined", "lockstep")} if c.get("mlp") else None, c.get("error"))
except Exception as e:
    print("bench: no JSON line (", e, ")")
PY
            ;;
    trace)  ( cd /tmp && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${n}_${label:-trace}" -o run -- python "$R/bench.py" $arg ) > "$O/${n}_${label:-trace}.log" 2>&1; echo "rc=$?" >> "$O/${n}_${label:-trace}.log" ;;
    pmc)    ctr=${arg%%:*}; barg=${arg#*:}
            ( cd /tmp && timeout $T rocprofv3 --pmc $ctr --output-format csv -d "$O/${n}_${label:-pmc}" -o run -- python "$R/bench.py" $barg ) > "$O/${n}_${label:-pmc}.log" 2>&1; echo "rc=$? counters: $ctr" >> "$O/${n}_${label:-pmc}.log" ;;
    py)     timeout $T python $arg > "$O/${n}_${label:-py}.log" 2>&1; echo "rc=$?" >> "$O/${n}_${label:-py}.log"; tail -5 "$O/${n}_${label:-py}.log" ;;
    sh)     timeout $T bash -c "$arg" > "$O/${n}_${label:-sh}.log" 2>&1; echo "rc=$?" >> "$O/${n}_${label:-sh}.log"; tail -5 "$O/${n}_${label:-sh}.log" ;;
    ab)     reps=${arg%%:*}; rest=${arg#*:}; libs=${rest%%:*}; barg=${rest#*:}
            : > "$O/${n}_${label:-ab}.txt"
            for r in $(seq 1 $reps); do for lib in ${libs//,/ }; do
              v=$(RSB_LIB_PATH=$R/raisimlib_amd/lib/$lib timeout $T python bench.py $barg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), round((d.get('lockstep') or {}).get('value') or 0))" 2>/dev/null)
              echo "$lib $v" >> "$O/${n}_${label:-ab}.txt"
            done; done; cat "$O/${n}_${label:-ab}.txt" ;;
    *)      echo "lease.sh: unknown step kind '$kind'" ;;
  esac
  echo "[lease $TAG] step $n $kind done in $(( $(date +%s) - t0 )) s"
done
