#!/bin/bash
# per-dispatch timeline of one 64-hypothesis batch (rocprofv3 --kernel-trace): kernel, grid, start offset, duration
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-trace}; mkdir -p $OUT
rm -rf /tmp/bt && rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $R/tools/batch_rate.py --scans 1 --hyp ${2:-64} > /tmp/bt.log 2>&1
f=$(find /tmp/bt -name "*kernel_trace.csv" | head -1)
python - $f <<'PY' | tee $OUT/batch_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last batch: from the last scan_keys_kernel<true> on
idx = [i for i, r in enumerate(rows) if "scan_keys_kernel<true>" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void soicp::", "")[:34]
    grid = "x".join(str(int(r[k]) // max(int(r[w]), 1)) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y")))
    print(f"{name:36s} wgs {grid:10s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us")
    prev_end = e
print("total", (prev_end - t0) / 1e3, "us")
PY
grep "batch mode" /tmp/bt.log
