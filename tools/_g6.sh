cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "batch" 2>&1 | tail -4
for rep in 1 2; do
for v in "" "SOICP_BATCH_ROUND0=near" "SOICP_BATCH_REPORT=full" "SOICP_BATCH_ROUND0=near SOICP_BATCH_REPORT=full"; do
  echo "== [$v]"; env $v python tools/batch_rate.py 2>&1 | grep "batch mode"
done; done
