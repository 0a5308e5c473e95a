#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bt
rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $R/tools/batch_rate.py --scans 3 > $O/bt.log 2>&1
f=$(find /tmp/bt -name "*kernel_trace.csv" | head -1)
python - $f > $O/batch_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if "solve_kernel" in n or "knn_plane" in n or "bin_" in n or "scan_keys" in n:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None: t0 = s
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>7s}x{r['Grid_Size_Y']:>3s} wg {r['Workgroup_Size_X']} lds {r['LDS_Block_Size']} vgpr {r.get('VGPR_Count','?')} agpr {r.get('Accum_VGPR_Count','?')} scr {r['Scratch_Size']} {n[:60]}")
PY
rm -rf /tmp/bp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d /tmp/bp -- python $R/tools/batch_rate.py --scans 2 > $O/bp.log 2>&1
python - /tmp/bp > $O/batch_pmc.txt <<'PY'
import sys, glob, csv, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "knn_plane" in k or "solve_kernel" in k:
            vals[(k.split("(")[0][:44], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    print(f"{k:46s} {c:22s} n {len(v):3d}  " + " ".join(f"{x:.3g}" for x in v[:10]))
PY
rm -rf /tmp/bp2
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/bp2 -- python $R/tools/batch_rate.py --scans 2 > $O/bp2.log 2>&1
python - /tmp/bp2 > $O/batch_pmc2.txt <<'PY'
import sys, glob, csv, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "knn_plane" in k or "solve_kernel" in k:
            vals[(k.split("(")[0][:44], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    print(f"{k:46s} {c:22s} n {len(v):3d}  " + " ".join(f"{x:.3g}" for x in v[:10]))
PY
tail -3 $O/bp2.log
