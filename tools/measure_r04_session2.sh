#!/bin/bash
# Round 4, second session: numbers of record of the final tree, one gpurun call.  usage: bash tools/measure_r04_session2.sh <tag>
# (the registration kernels -- kernels.hip -- did not change in this session: the PMC files and phase stamps of profiles/r04/ stand)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04s2}
cd $R; O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/pytest_gpu.txt
python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json
for r in 1 2 3; do python bench.py --steps 20 --warmup 5 2>> $O/bench.err | tail -1 > $O/bench_line_steps20_$r.json; done
python - $O <<'PY'
import json, sys
for name in ("bench_line", "bench_line_steps20_1", "bench_line_steps20_2", "bench_line_steps20_3"):
    d = json.load(open(f"{sys.argv[1]}/{name}.json"))
    loc = d.get("localization") or {}
    print(name, "value", round(d["value"], 1), "resident", round(d["entry_points"]["resident"], 1), "batch64", round(d["batch64"]["value"]),
          "knn us", round(1e3 * d["roofline"]["avg_launch_ms"], 2), "frac", round(d["roofline"]["frac"], 4),
          "| Localization raw / node order ms per frame", round(loc.get("raw_sweep_ms_per_frame", 0), 4), round(loc.get("node_order_ms_per_frame", 0), 4),
          "| cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", d.get("parity_vs_oracle_m_rad"))
PY
bash tools/prof_stats.sh $TAG 2>&1 | tail -24 | tee $O/prof_stats.txt
cp gpurun_out/prof_$TAG/*.csv $O/ 2>/dev/null
timeout 400 python tools/localization_rate.py --calls 64 --modes default,staged,hostbuilt,staged_hostbuilt,node,node_pageable,node_samequeue,node_hostbuilt,node_r3,default,staged,node 2>&1 | tail -14 | tee $O/localization_rate.txt
timeout 300 bash tools/localization_timeline.sh $TAG default 2>&1 | tail -3
timeout 300 bash tools/localization_timeline.sh $TAG staged 2>&1 | tail -3
timeout 300 bash tools/localization_timeline.sh $TAG node 2>&1 | tail -3
timeout 300 bash tools/prof_localization.sh $TAG 2>&1 | tail -3; cp gpurun_out/prof_$TAG/localization_kernel_stats.csv $O/ 2>/dev/null
timeout 300 python tools/f4_rates.py 2>&1 | tail -3 | tee $O/f4_rates.txt
timeout 500 python tools/soak_map_insert.py --oracle --seconds 40 2>&1 | tail -2 | tee $O/soak.txt
timeout 500 python tools/soak_localization.py --seconds 40 2>&1 | tail -2 | tee -a $O/soak.txt
timeout 300 python tools/soak_prefilter.py --seconds 30 2>&1 | tail -2 | tee -a $O/soak.txt
