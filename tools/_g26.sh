#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g26; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest.txt
timeout 100 python tools/soak_batch.py --seconds 60 --seed 77 2>&1 | tail -3 | tee $O/soak_batch.txt
timeout 100 python tools/soak_registration.py --seconds 45 --seed 78 2>&1 | tail -2 | tee $O/soak_registration.txt
timeout 100 python tools/soak_shards.py --seconds 30 --seed 79 2>&1 | tail -2 | tee $O/soak_shards.txt
python tools/batch_rate.py 2>&1 | grep "batch mode" | tee $O/batch_rate.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/bt
rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $R/tools/batch_rate.py --scans 2 > $O/bt.log 2>&1
f=$(find /tmp/bt -name "*kernel_trace.csv" | head -1)
python - $f > $O/batch_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_e = None
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_e) / 1e3 if prev_e else 0
    prev_e = e
    print(f"gap {gap:9.1f} dur {(e - s) / 1e3:8.1f} us grid {int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1):>5d}x{r['Grid_Size_Y']:>3s} {n[:50]}")
PY
tail -60 $O/batch_trace.txt
