#!/bin/bash
# round-4 soaks, one gpurun call; results under gpurun_out/soak_r04/
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/soak_r04; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 200 python tools/soak_registration.py --seconds 150 --seed 4 2>&1 | tail -3 > $O/soak_registration.txt
timeout 200 python tools/soak_registration.py --seconds 90 --seed 41 --scenes os1_128_2m 2>&1 | tail -3 > $O/soak_registration_headline.txt
timeout 200 python tools/soak_batch.py --seconds 150 --seed 4 2>&1 | tail -5 > $O/soak_batch.txt
timeout 200 python tools/soak_knn.py --seconds 100 --seed 4 2>&1 | tail -3 > $O/soak_knn.txt
timeout 200 python tools/soak_localization.py --seconds 120 --seed 4 2>&1 | tail -3 > $O/soak_localization.txt
timeout 200 python tools/soak_map_insert.py --seconds 80 --seed 12 --oracle 2>&1 | tail -2 > $O/soak_map_insert.txt
timeout 200 python tools/soak_shards.py --seconds 80 --seed 4 2>&1 | tail -3 > $O/soak_shards.txt
timeout 120 python tools/soak_prefilter.py --seconds 60 --seed 4 2>&1 | tail -2 > $O/soak_prefilter.txt
for f in $O/soak_*.txt; do echo "== $f"; cat $f; done
