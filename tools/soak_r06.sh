#!/bin/bash
# round-6 soaks on the final tree, one gpurun call (~6 min on the box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/soak_r06; mkdir -p $O
timeout 200 python tools/soak_registration.py --seconds 50 --seed 306 2>&1 | tail -1 > $O/soak_registration.txt
timeout 200 python tools/soak_registration.py --seconds 30 --seed 341 --scenes os1_128_2m 2>&1 | tail -1 > $O/soak_registration_headline.txt
timeout 300 python tools/soak_sequence.py --calls 200 --seed 306 2>&1 | tail -2 > $O/soak_sequence.txt; timeout 200 python tools/soak_sequence.py --calls 40 --scene os1_128_2m --seed 307 2>&1 | tail -2 >> $O/soak_sequence.txt
timeout 200 python tools/soak_binned_ahead.py --seconds 40 --seed 306 2>&1 | tail -1 > $O/soak_binned_ahead.txt
timeout 200 python tools/soak_batch.py --seconds 40 --seed 306 2>&1 | tail -1 > $O/soak_batch.txt
timeout 200 python tools/soak_knn.py --seconds 25 --seed 306 2>&1 | tail -1 > $O/soak_knn.txt
timeout 200 python tools/soak_localization.py --seconds 40 --seed 306 2>&1 | tail -1 > $O/soak_localization.txt
timeout 200 python tools/soak_map_insert.py --seconds 25 --seed 312 --oracle 2>&1 | tail -1 > $O/soak_map_insert.txt
timeout 200 python tools/soak_shards.py --seconds 30 --seed 306 2>&1 | tail -1 > $O/soak_shards.txt
timeout 200 python tools/soak_prefilter.py --seconds 40 --seed 306 2>&1 | tail -1 > $O/soak_prefilter.txt
timeout 200 python tools/soak_deskew.py --seconds 15 --seed 306 2>&1 | tail -1 > $O/soak_deskew.txt
for f in $O/soak_*.txt; do echo "== $(basename $f)"; cat $f; done | tee $O/all.txt
