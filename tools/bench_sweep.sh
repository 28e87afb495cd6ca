#!/bin/bash
# bench.py --no-cpu over a list of extra-argument strings; prints one short line per run
R=${GRAFT_REPO_ROOT:-$(pwd)}
for args in "$@"; do
  python $R/bench.py --no-cpu $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$args |', '%.1fM env-steps/s' % (d['value'] / 1e6), 'ms/step %.4f' % d['ms_per_step'], 'kernel %.1f us' % (1e3 * d['roofline']['kernel_ms_mean']), 'host enqueue %.4f ms/step' % d.get('host_enqueue_ms_per_step', -1), d['state_at_end'])"
done
