#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g37; mkdir -p $O
cd $R
python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json
python bench.py --steps 20 --warmup 5 2>> $O/bench.err | tail -1 > $O/bench_line_steps20.json
python - $O <<'PY'
import json, sys
for f in ("bench_line.json", "bench_line_steps20.json"):
    d = json.load(open(sys.argv[1] + "/" + f))
    print(f, "value", round(d["value"], 1), {k: round(v, 1) for k, v in d["entry_points"].items() if k != "note"}, "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4),
          "traffic", d["roofline"]["traffic"], "valu", (d["roofline"].get("valu_issue") or {}).get("valu_issue_frac"), "batch64", round(d["batch64"]["value"]), "overhead", d["host"]["fixed_overhead_ms_per_step"], "c_abi", d["host"]["c_abi_ms_per_step"])
PY
bash tools/two_rank_one_gpu.sh 2 map 2>&1 | tail -1 > $O/ranks2_map.json
bash tools/two_rank_one_gpu.sh 2 queries 2>&1 | tail -1 > $O/ranks2_queries.json
python - $O <<'PY'
import json, sys
for f in ("ranks2_map.json", "ranks2_queries.json"):
    try:
        d = json.load(open(sys.argv[1] + "/" + f)); print(f, "value", round(d["value"], 1), d["config"].get("shard_mode"), d["config"].get("peer_exchange"), (d.get("other_shard_mode") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
PY
