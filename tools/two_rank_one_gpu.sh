#!/bin/bash
# development aid: bench.py with N ranks on ONE GPU (peer exchange only; RCCL refuses two ranks per device).
# usage: bash tools/two_rank_one_gpu.sh [N=2] [map|queries]   -- N x SOICP_SOLVE_WORKGROUPS must stay <= the device's compute units
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}
MODE=${2:-map}
export SOICP_BENCH_DEVICE=0 SOICP_BENCH_NO_RCCL=1 SOICP_SOLVE_WORKGROUPS=$((200 / N))
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $N --steps 24 --warmup 4 --no-cpu-baseline --shard-mode $MODE 2> gpurun_out/ranks$N.err | tail -1 > gpurun_out/ranks$N.json
grep -v "hostname of the client\|amdgpu.ids\|^$" gpurun_out/ranks$N.err | tail -5
python - <<PY
import json
d=json.load(open("gpurun_out/ranks$N.json"))
print("N=$N value %.0f reg/s (%.3f ms) peer_exchange=%s | %s" % (d["value"], d["ms_per_step"], d["config"]["peer_exchange"], d["config"]["parallelism"][:60]))
print("   shard_mode", d["config"]["shard_mode"], "| other mode:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in (d.get("other_shard_mode") or {}).items()})
print("   predicted_scaling query_split", {k: round(v) for k, v in ((d.get("predicted_scaling") or {}).get("query_split") or {}).items()})
print("   entry_points", {k: round(v) for k, v in d["entry_points"].items() if k != "note"}, "| batch64 %.0f (%d per rank) | executed %s" % (d["batch64"]["value"], d["batch64"]["hypotheses_per_rank"], d["executed"]))
PY
