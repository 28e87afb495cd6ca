#!/bin/bash
# Round-3 GPU call K: group-local sweep in the large-contact kernel classes (kmax > 8): parity tests, then config 5 against the
# previous library (_ab_prev/librsb.so = per-pass exchange) on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q -k "atlas or config5 or 16_contact or fuzz or multi_step or population or kat" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for regime in standing collapsing; do
  for i in 1 2; do
    for v in librsb.so librsb.prev.so; do
      RSB_LIB_PATH=$R/raisimlib_amd/lib/$v timeout 300 python $R/bench.py --config 5 --atlas-regime $regime --no-cpu --steps 150 --warmup 30 2>$O/err.txt | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('config 5 %-10s %-16s %.2f M  kernel %.4f ms  sweeps mean %.1f' % ('$regime', '$v', b['value']/1e6, b['roofline']['kernel_ms_mean'], b['state_at_end']['solver_iters_mean']))" 2>&1 | tee -a $O/ab_glocal.txt
    done
  done
done
timeout 200 python $R/bench.py --config 5 --no-cpu --steps 150 --warmup 30 --target-amplitude 0.25 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('config 5 standing, target noise x0.25: %.2f M  kernel %.4f ms  sweeps mean %.1f max %d contacts %.2f' % (b['value']/1e6, b['roofline']['kernel_ms_mean'], b['state_at_end']['solver_iters_mean'], b['state_at_end']['solver_iters_max'], b['state_at_end']['contacts_per_env']))" | tee -a $O/ab_glocal.txt
