#!/bin/bash
# development aid: bench.py with 2 ranks on ONE GPU (peer exchange only; RCCL refuses two ranks per device)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SOICP_BENCH_DEVICE=0 SOICP_BENCH_NO_RCCL=1 SOICP_SOLVE_WORKGROUPS=100
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 24 --warmup 4 --no-cpu-baseline 2> gpurun_out/two_rank.err | tail -1 > gpurun_out/two_rank.json
tail -5 gpurun_out/two_rank.err
python - <<PY
import json
d=json.load(open("gpurun_out/two_rank.json"))
print(d["value"], d["ms_per_step"], d["config"]["parallelism"], d["config"]["peer_exchange"], d["entry_points"], d["kernels"], d["batch64"]["value"], d["executed"])
PY
