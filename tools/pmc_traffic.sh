#!/bin/bash
# HBM traffic of knn_plane_kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC has 4 slots:
# FETCH_SIZE costs 3, WRITE_SIZE 2), corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
#   read bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE tallies 128-B requests at 64 B), write bytes = WRITE_SIZE * 1024.
# usage (GPU box): bash tools/pmc_traffic.sh <tag>   -> gpurun_out/pmc_<tag>/{p_FETCH_SIZE.csv,p_WRITE_SIZE.csv,knn_traffic.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -- \
    python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary --entry resident > /tmp/pmc_$ctr.log 2>&1
done
python - $OUT <<'PY'
import sys, glob, csv, collections, json, hashlib, os
out = sys.argv[1]
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sha = hashlib.sha256(open(R + "/superodom_amd/csrc/kernels.hip", "rb").read()).hexdigest()  # bench.py quotes the traffic only for the same kernel source
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pmc_{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    with open(f"{out}/p_{ctr}.csv", "w") as fh:
        fh.write("kernel,counter,launches,mean_per_launch,min,max\n")
        for k, v in sorted(acc.items()):
            fh.write(f'"{k}",{ctr},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}\n')
    # launches that did real work only (no-op launches after convergence read a few hundred bytes)
    v = [x for k, vs in acc.items() if "knn_plane" in k for x in vs]
    real = [x for x in v if x > 0.2 * max(v)] if v else []
    res[ctr] = (sum(real) / len(real)) if real else None
    res[ctr + "_launches"] = len(real)
    v = [x for k, vs in acc.items() if "solve_kernel" in k for x in vs]
    real = [x for x in v if x > 0.2 * max(v)] if v else []
    res["solve_" + ctr] = (sum(real) / len(real)) if real else None
j = {"kernel": "soicp::knn_plane_kernel", "round": 6, "kernels_hip_sha256": sha,
     "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, bench.py --steps 4 --warmup 1); no-op launches excluded",
     "FETCH_SIZE_KB_per_launch": res["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": res["WRITE_SIZE"],
     "launches": [res["FETCH_SIZE_launches"], res["WRITE_SIZE_launches"]],
     "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> read bytes = 2 * FETCH_SIZE * 1024",
     "hbm_bytes_per_launch": int(2 * res["FETCH_SIZE"] * 1024 + res["WRITE_SIZE"] * 1024) if res["FETCH_SIZE"] and res["WRITE_SIZE"] else None,
     "solve_kernel": {"FETCH_SIZE_KB_per_launch": res["solve_FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": res["solve_WRITE_SIZE"],
                      "hbm_bytes_per_launch": int(2 * res["solve_FETCH_SIZE"] * 1024 + res["solve_WRITE_SIZE"] * 1024) if res["solve_FETCH_SIZE"] and res["solve_WRITE_SIZE"] else None}}
json.dump(j, open(f"{out}/knn_traffic.json", "w"), indent=1)
print(json.dumps(j))
PY
