cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in default 0; do
    if [ $v = default ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=0; fi
    python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('HSA_ENABLE_INTERRUPT=$v', 'value %.1f spread %.3f' % (d['value']/1e6, d['spread_rel']), 'pipelined %.1f %.3f' % (d['pipelined']['value']/1e6, d['pipelined']['spread_rel']), 'lockstep %.1f %.3f' % (d['lockstep']['value']/1e6, d['lockstep']['spread_rel']))"
  done
done
