# A/B of two prebuilt library variants on ONE box for the batched path: superodom_amd/lib/libsoicp_base.so and libsoicp_wavectl.so (built in the
# development container from the two trees and left beside libsoicp.so; they travel with the snapshot): batch of 64 and of 8, three rounds, then the batch tests
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for v in base wavectl; do cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so; echo "$v: $(python tools/batch_rate.py --hyp 64 --scans 3 | tail -1 | cut -c1-120) | $(python tools/batch_rate.py --hyp 8 --scans 8 | tail -1 | cut -c30-110)"; done; done
cp superodom_amd/lib/libsoicp_wavectl.so superodom_amd/lib/libsoicp.so
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -2
