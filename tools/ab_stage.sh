cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "steady 1" "cold 1" "steady 2" "steady 0"; do
  set -- $cfg
  SOICP_STAGE_AT=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-open-scene --no-concurrent --stage-protocol $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 at=$2', 'value %.0f resident %.0f ratio %.3f host %.0f' % (d['value'], d['entry_points']['resident'], d['value']/d['entry_points']['resident'], d['entry_points']['host']))"
done; done
python bench.py --steps 240 --warmup 8 --no-cpu-baseline --no-stock --no-open-scene --no-concurrent 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steady K=240', 'value %.0f resident %.0f ratio %.3f' % (d['value'], d['entry_points']['resident'], d['value']/d['entry_points']['resident']))"
