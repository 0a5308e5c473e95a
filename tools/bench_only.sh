#!/bin/bash
# quick A/B: bench line only (no CPU baseline).  usage: bash tools/bench_only.sh <tag> [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
cd $R; mkdir -p gpurun_out
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline "$@" 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_${TAG}_$rep.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_$rep.json"))
k=d["kernels"]
print("rep $rep", {a:round(b,1) for a,b in d["entry_points"].items() if a!="note"}, "knn_launch_us %.2f"%(1e3*d["roofline"]["avg_launch_ms"]), "knn %.1f solve %.1f bin %.1f rest %.1f"%(1e3*k["knn_ms_per_registration"],1e3*k["solve_ms_per_registration"],1e3*k["binning_ms_per_registration"],1e3*k["rest_ms_per_registration"]), "batch64 %.0f"%d["batch64"]["value"])
PY
done
