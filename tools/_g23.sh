#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g23; mkdir -p $O
cd $R
for h in 64 8 1; do python tools/batch_stamps.py --hyp $h --scans 2 2>&1 | tail -6; done | tee $O/stamps.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "64 auto" "64 1" "16 auto" "8 auto" "2 auto" "1 auto"; do
  set -- $cfg
  rm -rf /tmp/bt
  if [ "$2" = "auto" ]; then unset SOICP_BATCH_WG_PER_CU; else export SOICP_BATCH_WG_PER_CU=$2; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $R/tools/batch_rate.py --hyp $1 --scans 2 > $O/bt_$1_$2.log 2>&1
  f=$(find /tmp/bt -name "*kernel_trace.csv" | head -1)
  echo "== hyp $1 wg/cu $2" >> $O/traces.txt
  python - $f >> $O/traces.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "solve_kernel" in n or "knn_plane" in n:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"dur {(e - s) / 1e3:8.1f} us grid {int(r['Grid_Size_X'])//256:>5d}x{r['Grid_Size_Y']:>3s} {n[13:36]}")
PY
done
tail -80 $O/traces.txt
