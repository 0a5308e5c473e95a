#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g27; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "batch" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee $O/pytest.txt
python tools/batch_rate.py 2>&1 | grep "batch mode" | tee $O/batch_rate.txt
SOICP_BATCH_CHAIN=0 python tools/batch_rate.py 2>&1 | grep "batch mode" | tee -a $O/batch_rate.txt
python tools/batch_rate.py --scans 8 2>&1 | grep "batch mode" | tee -a $O/batch_rate.txt
timeout 100 python tools/soak_batch.py --seconds 40 --seed 91 2>&1 | tail -2 | tee $O/soak_batch.txt
