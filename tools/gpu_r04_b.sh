#!/bin/bash
# Round 4, call B: the GPU suite on the fused drop-in path, the template path's rate, the default bench line with `secondary`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
nproc > $O/nproc.txt
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
for t in 8 16 32; do RSB_FIBER_THREADS=$t timeout 200 python tools/bench_gym.py 4096 40 $t > $O/bench_gym_t$t.json 2>>$O/bench_gym.err; python -c "
import json; b=json.load(open('$O/bench_gym_t$t.json')); print('threads $t', '%.2fM env-steps/s %.3f ms/step launches %d' % (b['template_path']['env_steps_per_s']/1e6, b['template_path']['ms_per_control_step'], b['template_path']['launches']), ' device env %.1fM' % (b['device_env_host_buffers']['env_steps_per_s']/1e6))"; done
RSB_VIEW_FUSE=0 RSB_FIBER_THREADS=16 timeout 200 python tools/bench_gym.py 4096 40 16 > $O/bench_gym_nofuse.json 2>>$O/bench_gym.err; python -c "
import json; b=json.load(open('$O/bench_gym_nofuse.json')); print('no fuse', '%.2fM' % (b['template_path']['env_steps_per_s']/1e6))"
cd /tmp && export TMPDIR=/tmp
( time timeout 400 python $R/bench.py --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 > $O/bench_default.json ) 2>&1 | grep real
python - <<PY
import json
b=json.load(open("$O/bench_default.json"))
print("c2 %.2fM kernel %.4f" % (b["value"]/1e6, b["roofline"]["kernel_ms_mean"]), "valu", b["roofline"]["valu_issue"] and {k: round(v, 3) for k, v in b["roofline"]["valu_issue"].items() if isinstance(v, float)})
for k, v in b.get("secondary", {}).items(): print(k, v.get("error") or "%.2fM kernel %.4f frac %.4f cpu %.2fM" % (v["value"]/1e6, v["kernel_ms_mean"], v["roofline"]["frac"], v["cpu_baseline"]["value"]/1e6))
print("template", b.get("boundary_template_path"))
PY
