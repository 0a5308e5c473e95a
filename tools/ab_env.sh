# usage: bash tools/ab_env.sh "VAR=val VAR2=val" ...   -- one bench line (driver protocol, secondary blocks off) per environment, twice, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-open-scene --no-concurrent 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('%-40s' % '$cfg', 'value %.0f resident %.0f ratio %.3f host %.0f chained %.0f | knn %.1f solve %.1f bin %.1f | batch64 %.0f' % (d['value'], d['entry_points']['resident'], d['value']/d['entry_points']['resident'], d['entry_points']['host'], d['entry_points'].get('chained', 0), 1e3*k['knn_ms_per_registration'], 1e3*k['solve_ms_per_registration'], 1e3*k['binning_ms_per_registration'], d['batch64']['value']))"
done; done
