#!/bin/bash
# SQ instruction counters of knn_plane_kernel / solve_kernel (one rocprofv3 --pmc pass per SOICP_ABLATE value).
# usage (GPU box): bash tools/pmc_knn.sh "0 [8 24 2 ...]"  -> gpurun_out/pmc_knn/ablate_<n>.txt and, for ablate 0,
#                  gpurun_out/pmc_knn/knn_counters.json (copied to profiles/knn_counters.json: bench.py reads it for valu_issue_frac)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_knn
mkdir -p $OUT
for a in ${1:-0}; do
  rm -rf /tmp/pmc_$a
  SOICP_ABLATE=$a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_$a -- \
    python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary --entry resident > /tmp/pmc_$a.log 2>&1
  python - "$a" /tmp/pmc_$a $OUT > $OUT/ablate_$a.txt <<'PY'
import sys, glob, csv, collections, json, hashlib, os
a, d, out = sys.argv[1], sys.argv[2], sys.argv[3]
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sha = hashlib.sha256(open(R + "/superodom_amd/csrc/kernels.hip", "rb").read()).hexdigest()  # bench.py quotes the counters only for the same kernel source
vals = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "knn_plane" not in k and "eval_kernel" not in k and "solve_kernel" not in k:
            continue
        vals[(k.split("(")[0][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
print("ablate", a, " (REAL launches only: a launch after convergence is a no-op with a few thousand instructions)")
real = {}
for (k, c), v in sorted(vals.items()):
    ref = vals[(k, "SQ_INSTS_VALU")]
    keep = [x for x, r in zip(v, ref) if r > 0.2 * max(ref)] if len(ref) == len(v) else v
    real[(k, c)] = sum(keep) / max(len(keep), 1)
    print(f"{k:42s} {c:22s} launches {len(v):4d} real {len(keep):4d} mean(real) {real[(k, c)]:14.1f}")
if a == "0":
    kk = [k for (k, c) in real if "knn_plane" in k][0]
    j = {"kernel": "soicp::knn_plane_kernel", "round": 6, "kernels_hip_sha256": sha,
         "source": "tools/pmc_knn.sh: rocprofv3 --pmc SQ_INSTS_VALU ... --kernel-trace (own pass, no other trace domain), bench.py --steps 4 --warmup 1 --entry resident; no-op launches excluded",
         "valu_wave_insts_per_launch": real[(kk, "SQ_INSTS_VALU")], "salu_wave_insts_per_launch": real[(kk, "SQ_INSTS_SALU")],
         "lds_wave_insts_per_launch": real[(kk, "SQ_INSTS_LDS")], "waves_per_launch": real[(kk, "SQ_WAVES")],
         "wave_cycles_per_launch_x4": real[(kk, "SQ_WAVE_CYCLES")], "active_inst_valu_x4": real[(kk, "SQ_ACTIVE_INST_VALU")], "shader_clock_ghz": 2.15}
    ks = [k for (k, c) in real if "solve_kernel" in k]
    if ks:
        j["solve_kernel"] = {"valu_wave_insts_per_launch": real[(ks[0], "SQ_INSTS_VALU")], "active_inst_valu_x4": real[(ks[0], "SQ_ACTIVE_INST_VALU")],
                             "wave_cycles_x4": real[(ks[0], "SQ_WAVE_CYCLES")], "waves": real[(ks[0], "SQ_WAVES")]}
    json.dump(j, open(out + "/knn_counters.json", "w"), indent=1)
PY
  cat $OUT/ablate_$a.txt
done
