cd $GRAFT_REPO_ROOT
for m in map queries; do for n in 2 4; do echo "== N=$n mode $m"; bash tools/two_rank_one_gpu.sh $n $m 2>&1 | tail -6; cp gpurun_out/ranks$n.json gpurun_out/ranks${n}_$m.json; done; done
echo "== N=1 default line"; timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/b1.err | tail -1 > gpurun_out/b1.json; python -c "
import json; d=json.load(open('gpurun_out/b1.json')); print(round(d['value'],1), d['entry_points'], d['host']); print(json.dumps(d['predicted_scaling'])[:1500])"
