#!/bin/bash
# round-3 soaks + N ranks on one GPU, one gpurun call; results under gpurun_out/soak_r03/
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/soak_r03; mkdir -p $O
timeout 260 python tools/soak_registration.py --seconds 200 --seed 3 2>&1 | tail -3 > $O/soak_registration.txt
timeout 260 python tools/soak_batch.py --seconds 200 --seed 3 2>&1 | tail -5 > $O/soak_batch.txt
timeout 200 python tools/soak_map_insert.py --seconds 120 --seed 11 --oracle 2>&1 | tail -2 > $O/soak_map_insert.txt
for N in 2 4; do bash tools/two_rank_one_gpu.sh $N > $O/ranks$N.txt 2>&1; cp gpurun_out/ranks$N.json $O/; done
cat $O/*.txt
