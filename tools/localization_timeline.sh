#!/bin/bash
# timeline of ONE Localization() call (registration + map insert): kernels and copies with start offsets, durations, gaps
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
# usage: bash tools/localization_timeline.sh <tag> [mode of tools/localization_rate.py, default "default"]
OUT=$R/gpurun_out/${1:-loc_tl}; mkdir -p $OUT
MODE=${2:-default}
rm -rf /tmp/ltl && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ltl -- python $R/tools/localization_rate.py --modes $MODE > /tmp/ltl.log 2>&1
python - <<'PY' | tee $OUT/localization_timeline_$MODE.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/ltl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void soicp::", "").replace("soicp::", "")[:44]))
for f in glob.glob("/tmp/ltl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))[:38]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("scan_keys_kernel")]
a, b = idx[-3], idx[-2]
t0 = rows[a][0]; prev_end = t0; busy = 0
for s, e, name in rows[a:b]:
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}")
    busy += e - s; prev_end = max(prev_end, e)
print("period (scan_keys to scan_keys) %.1f us, busy %.1f us" % ((rows[b][0] - t0) / 1e3, busy / 1e3))
PY
tail -2 /tmp/ltl.log
