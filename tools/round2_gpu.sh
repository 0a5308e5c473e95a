#!/bin/bash
# One gpurun call of round 2: GPU test suite, smoke, bench line, rocprofv3 kernel statistics.
# usage (GPU box): bash tools/round2_gpu.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
shift
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=12 "$@" 2>&1 | tail -60 ) > gpurun_out/pytest_$TAG.log
tail -25 gpurun_out/pytest_$TAG.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/smoke_$TAG.log; cat gpurun_out/smoke_$TAG.log
timeout 600 python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print({k:d[k] for k in ("value","ms_per_step","entry_points")}); print(d["roofline"]["avg_launch_ms"], d["kernels"]); print(d.get("batch64")); print({k:v.get("value") for k,v in d.items() if k.startswith("cpu_baseline")}, d.get("parity_vs_oracle_m_rad"))
PY
