#!/bin/bash
# Round-4 evidence run (through gpurun): full GPU test-suite + smoke, the default bench line (headline + secondary configs + template path), bench
# lines of configs 2 / 3 / 5 on their own, rocprofv3 kernel traces, PMC passes (one counter group per pass, never combined with trace
# domains), wave / phase diagnostics, the drop-in path's thread sweep.
# Writes under gpurun_out/<tag>/; tools/summarise_r03.py <tag> turns it into the tracked profiles/<tag>_* files.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
nproc > $O/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -14 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
( time timeout 400 python $R/bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json ) 2> $O/bench_default.time
( time timeout 300 python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_c2_driverlike.json ) 2> $O/bench_driverlike.time
timeout 400 python $R/bench.py --no-secondary 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
timeout 400 python $R/bench.py --config 3 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
timeout 400 python $R/bench.py --config 3 --per-env-maps --no-cpu 2>$O/bench_c3pe.err | tail -1 > $O/bench_c3_per_env_maps.json
timeout 500 python $R/bench.py --config 5 2>$O/bench_c5.err | tail -1 > $O/bench_c5.json
timeout 500 python $R/bench.py --config 5 --atlas-regime collapsing 2>$O/bench_c5c.err | tail -1 > $O/bench_c5_collapsing.json
for c in 2 3 5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c$c -o bench -- python $R/bench.py --no-cpu --config $c > $O/trace_c$c.log 2>&1
done
for c in 2 3 5; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c$c -o run -- python $R/bench.py --no-cpu --config $c --steps 50 --warmup 50 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c$c -o run -- python $R/bench.py --no-cpu --config $c --steps 50 --warmup 50 > /dev/null 2>&1
done
# the SQ passes characterise the kernel itself: --lockstep (the plain kernel class; a pipelined wave's cycles include its wait for a predecessor)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o run -- python $R/bench.py --no-cpu --lockstep --steps 50 --warmup 50 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o run -- python $R/bench.py --no-cpu --lockstep --steps 50 --warmup 50 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sq_c5 -o run -- python $R/bench.py --no-cpu --lockstep --config 5 --steps 30 --warmup 30 > /dev/null 2>&1
timeout 300 python $R/bench.py --no-cpu --no-secondary --lockstep 2>/dev/null | tail -1 > $O/bench_c2_lockstep.json
timeout 300 python $R/bench.py --no-cpu --no-secondary --force-collective 2>/dev/null | tail -1 > $O/bench_c2_forced_collective.json
timeout 300 python $R/bench.py --no-cpu --no-secondary --force-collective --lockstep 2>/dev/null | tail -1 > $O/bench_c2_forced_collective_lockstep.json
timeout 300 python $R/tools/exp/pipeline.py > $O/exp_pipeline.txt 2>&1
timeout 200 python $R/tools/diag_phases.py > $O/diag_phases.txt 2>&1
RSB_PROF_FINE=1 timeout 200 python $R/tools/diag_waves.py > $O/diag_waves.txt 2>&1
timeout 300 python $R/tools/diag_atlas_phases.py standing > $O/diag_atlas.txt 2>&1
timeout 200 python $R/tools/bench_vecenv.py > $O/bench_vecenv.txt 2>&1
for t in 1 8 16 32 64 128; do RSB_FIBER_THREADS=$t timeout 300 python $R/tools/bench_gym.py 4096 1000 $t > $O/bench_gym_t$t.json 2>>$O/bench_gym.err; done
cp $O/bench_gym_t32.json $O/bench_gym.json
RSB_VIEW_FUSE=0 RSB_FIBER_THREADS=32 timeout 300 python $R/tools/bench_gym.py 4096 1000 32 > $O/bench_gym_t32_nofuse.json 2>>$O/bench_gym.err
python - <<PY
import json
for n in ("c2","c2_driverlike","c2_lockstep","c2_forced_collective","c2_forced_collective_lockstep","c3","c3_per_env_maps","c5","c5_collapsing"):
    try:
        b=json.load(open("$O/bench_%s.json"%n)); r=b["roofline"]
        print(n, "%.2fM"%(b["value"]/1e6), "lockstep %.2fM" % ((b.get("lockstep") or {}).get("value", 0)/1e6), "ms/step %.4f"%b["ms_per_step"], "kernel %.4f max %.4f n %d"%(r["kernel_ms_mean"], r["kernel_ms_max"], r["kernel_launches_timed"]), b["config"]["regime"], b.get("cpu_baseline",{}).get("value"), b.get("cpu_baseline",{}).get("cores"))
    except Exception as e: print(n, "FAILED", e)
b=json.load(open("$O/bench_default.json"))
print("default line: c2 %.2fM" % (b["value"]/1e6), {k: (v.get("error") or round(v["value"]/1e6, 2)) for k, v in b["secondary"].items()}, "template %.2fM @ %d threads" % (b["boundary_template_path"]["env_steps_per_s"]/1e6, b["boundary_template_path"]["host_threads"]))
print(open("$O/bench_default.time").read().strip().splitlines()[0] if open("$O/bench_default.time").read().strip() else "")
for t in (1, 8, 16, 32, 64, 128):
    try:
        g=json.load(open("$O/bench_gym_t%d.json"%t)); print("gym threads", t, "%.2fM env-steps/s %.3f ms/step" % (g["template_path"]["env_steps_per_s"]/1e6, g["template_path"]["ms_per_control_step"]))
    except Exception as e: print("gym", t, "FAILED", e)
g=json.load(open("$O/bench_gym_t32_nofuse.json")); print("gym threads 32, no fuse", "%.2fM" % (g["template_path"]["env_steps_per_s"]/1e6))
PY
ls $O
