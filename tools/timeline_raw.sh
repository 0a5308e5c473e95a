cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tl_r
rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl_r -- python $R/bench.py --steps 24 --warmup 4 --entry resident --no-cpu-baseline --no-profile-pass --no-secondary --no-kernel-events > /tmp/tl_r.log 2>&1
ls /tmp/tl_r/*/ | head
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl_r/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_keys" in r["Kernel_Name"]]
h = glob.glob("/tmp/tl_r/**/*hip_api_trace.csv", recursive=True)
api = sorted(csv.DictReader(open(h[0])), key=lambda r: int(r["Start_Timestamp"])) if h else []
for k in (-6, -4):
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]["Start_Timestamp"]); prev = t0
    print("---- registration", k)
    ev = []
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, "K  %-30s dur %6.2f gap %5.2f" % (r["Kernel_Name"].split("(")[0].replace("void soicp::", "")[:30], (e - s) / 1e3, (s - prev) / 1e3)))
        prev = e
    t1 = int(rows[b]["Start_Timestamp"])
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 - 20000 <= s <= t1:
            ev.append((s, "   api %-28s %6.2f us" % (r["Function"][:28], (e - s) / 1e3)))
    for s, txt in sorted(ev):
        print("%9.2f  %s" % ((s - t0) / 1e3, txt))
PY
