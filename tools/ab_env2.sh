# usage: bash tools/ab_env2.sh "VAR=val" ...   -- bench lines (driver protocol, no secondary blocks, no CPU baseline) per environment, three times, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s' % '$cfg', 'value %.0f ms %.4f c_abi %.4f knn us %.2f' % (d['value'], d['ms_per_step'], d['host']['c_abi_ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"
done; done
