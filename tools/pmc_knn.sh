#!/bin/bash
# SQ instruction counters of knn_plane_kernel under the SOICP_ABLATE switches (one rocprofv3 --pmc pass each).
# usage (GPU box): bash tools/pmc_knn.sh "0 8 24 2"   -> gpurun_out/pmc_knn/ablate_<n>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_knn
mkdir -p $OUT
for a in ${1:-0}; do
  rm -rf /tmp/pmc_$a
  SOICP_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_$a -- \
    python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary > /tmp/pmc_$a.log 2>&1
  python - "$a" /tmp/pmc_$a > $OUT/ablate_$a.txt <<'PY'
import sys, glob, csv, collections
a, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "knn_plane" not in k and "eval_kernel" not in k and "solve_kernel" not in k:
            continue
        key = (k.split("(")[0][:40], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
print("ablate", a)
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:42s} {c:22s} launches {n:4d} mean {v / n:14.1f}")
PY
  cat $OUT/ablate_$a.txt
done
