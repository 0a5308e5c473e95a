# usage: bash tools/ab_stock.sh "VAR=val" ...  -- the stock-operating-point block of the bench line per environment, twice, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-open-scene --no-concurrent 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stock']
print('%-32s' % '$cfg', 'value %.0f |' % d['value'], ' | '.join('%s: reg %.4f ms frame %.4f ms q %d outer %.2f lm %.2f' % (k, v['registration_ms'], v['node_frame_ms'], v['sampled_queries'], v['outer_iterations'], v['lm_iterations']) for k, v in s.items() if 'registration_ms' in v))"
done; done
