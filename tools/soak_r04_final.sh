#!/bin/bash
# round-4 soaks on the final tree (after the batched-solve rework, chained rounds, binned records), one gpurun call
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/soak_r04_final; mkdir -p $O
timeout 200 python tools/soak_registration.py --seconds 80 --seed 104 2>&1 | tail -1 > $O/soak_registration.txt
timeout 200 python tools/soak_registration.py --seconds 40 --seed 141 --scenes os1_128_2m 2>&1 | tail -1 > $O/soak_registration_headline.txt
timeout 200 python tools/soak_batch.py --seconds 100 --seed 104 2>&1 | tail -1 > $O/soak_batch.txt
timeout 200 python tools/soak_knn.py --seconds 40 --seed 104 2>&1 | tail -1 > $O/soak_knn.txt
timeout 200 python tools/soak_localization.py --seconds 60 --seed 104 2>&1 | tail -1 > $O/soak_localization.txt
timeout 200 python tools/soak_map_insert.py --seconds 30 --seed 112 --oracle 2>&1 | tail -1 > $O/soak_map_insert.txt
timeout 200 python tools/soak_shards.py --seconds 50 --seed 104 2>&1 | tail -1 > $O/soak_shards.txt
for f in $O/soak_*.txt; do echo "== $(basename $f)"; cat $f; done | tee $O/all.txt
