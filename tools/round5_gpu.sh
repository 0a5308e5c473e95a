#!/bin/bash
# Round 5 GPU session driver.  usage: bash tools/round5_gpu.sh <tag> [stages]
#   c  tests/test_gpu_conditioning.py (-s: prints cond(JtJ) and the deltas)
#   t  the whole -m gpu suite
#   b  one bench line, driver protocol (--steps 20 --warmup 5), every secondary block
#   g  bench.py --gpus 2 on this one-GPU box, started WITHOUT a launcher (first-run proofing of the N > 1 start)
#   s  in-kernel phase stamps of the solve / k-NN launches (instrumented rebuild; restores the production build afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
ST=${2:-ctbg}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
summ() { python - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def r(x, n=1):
    return round(x, n) if isinstance(x, (int, float)) else x
print("value %.1f ms/step %.4f | entry_points %s | knn us %.2f frac %.4f" % (d["value"], d["ms_per_step"], {k: r(v) for k, v in d["entry_points"].items() if k != "note"},
      1e3 * d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
print("  kernels", {k: r(v, 4) for k, v in (d.get("kernels") or {}).items() if k.endswith("_ms_per_registration")}, "| parity", d.get("parity_vs_oracle_m_rad"), d.get("parity_iteration_counts_and_histograms_equal"))
b = d.get("batch64") or {}
print("  batch64 %s by size %s" % (r(b.get("value")), {k: r(v) for k, v in (b.get("registrations_per_s_by_batch_size") or {}).items()}))
print("  predicted batch64", (d.get("predicted_scaling") or {}).get("batch64_replicated_map"), "8gpu/1gpu", r((d.get("predicted_scaling") or {}).get("batch64_8gpu_over_1gpu"), 2))
print("  concurrent", {k: (r(v.get("registrations_per_s")), v.get("stats_flags"), v.get("max_pose_delta_vs_timed_loop_m_or_rad"), v.get("error")) for k, v in (d.get("concurrent_contexts") or {}).items() if isinstance(v, dict)})
for k, v in (d.get("stock") or {}).items():
    print("  stock", k, {a: r(b_, 4) for a, b_ in v.items() if a not in ("config", "note")})
o = d.get("open_scene") or {}
print("  open_scene", {a: r(b_, 4) for a, b_ in o.items() if a not in ("workload", "pack_light")}, o.get("pack_light"))
print("  localization", {k: r(v, 4) for k, v in (d.get("localization") or {}).items() if k != "note"})
PY
}
if [[ $ST == *c* ]]; then
  timeout 600 python -m pytest tests/test_gpu_conditioning.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 | tee $O/pytest_conditioning.log
fi
if [[ $ST == *t* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -120 > $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
fi
if [[ $ST == *b* ]]; then
  T0=$SECONDS; timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_steps20.json
  echo "bench wall $((SECONDS - T0)) s"; tail -3 $O/bench.err
  summ $O/bench_steps20.json
fi
if [[ $ST == *g* ]]; then
  T0=$SECONDS; timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_gpus2.err > $O/bench_gpus2.out
  echo "wall $((SECONDS - T0)) s"; echo "rc=$? stdout lines: $(grep -c . $O/bench_gpus2.out)"; tail -4 $O/bench_gpus2.err
  python - $O/bench_gpus2.out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("N=2 on one device: value %.0f (%.3f ms) n_gpus %d peer %s shard_mode %s | other %s | batch64 %.0f" % (d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["peer_exchange"],
          d["config"]["shard_mode"], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in (d.get("other_shard_mode") or {}).items()}, (d.get("batch64") or {}).get("value", 0)))
except Exception as e:
    print("no JSON line:", e)
PY
fi
if [[ $ST == *p* ]]; then  # phase stamps of the production arithmetic (PROF instantiation, no controller stamps): fit + eval records
  SOICP_ABLATE=128 python tools/eval_stamps.py 2>&1 | grep "^fit\|^eval" | tee $O/phase_stamps_fit_eval.txt
fi
if [[ $ST == *s* ]]; then
  bash tools/lm_stamps.sh 2>&1 | tail -45 | tee $O/phase_stamps.txt
  python -m superodom_amd.build --force > /dev/null 2>&1
fi
