#!/bin/bash
# timeline of ONE node-order frame (so_icp_prefilter_scan of the raw sweep + Localization() of the filtered cloud incl. its insert): kernels and copies
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-node_tl}; mkdir -p $OUT
rm -rf /tmp/ntl && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ntl -- python $R/tools/localization_rate.py --modes node --calls 24 > /tmp/ntl.log 2>&1
python - <<'PY' | tee $OUT/node_timeline.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/ntl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s " % r.get("Queue_Id", "?") + r["Kernel_Name"].split("(")[0].replace("void soicp::", "").replace("soicp::", "")[:60]))
for f in glob.glob("/tmp/ntl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))[:38]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "insert_report_kernel" in r[2]]
a, b = idx[-4], idx[-3]
t0 = rows[a][1]; prev_end = t0; busy = 0
for s, e, name in rows[a + 1:b + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}")
    busy += e - s; prev_end = max(prev_end, e)
print("period (report to report) %.1f us, busy %.1f us" % ((rows[b][1] - t0) / 1e3, busy / 1e3))
PY
tail -2 /tmp/ntl.log
