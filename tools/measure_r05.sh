#!/bin/bash
# Round-5 numbers of record, one gpurun call.  usage: bash tools/measure_r05.sh <tag> [stages]
#   t  the -m gpu suite            b  bench lines (default + 2 x driver protocol)     k  rocprofv3 --kernel-trace --stats (with / without speculation)
#   m  PMC traffic + SQ counters (separate passes; tied to kernels.hip by sha256)      p  in-kernel phase stamps (PROF instantiation)
#   l  Localization() / node-order rates                                               g  bench.py --gpus 2 on this one-GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
ST=${2:-tbkmpl}
cd $R; O=gpurun_out/$TAG; mkdir -p $O
line() { python - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = lambda x, n=1: round(x, n) if isinstance(x, (int, float)) else x
print("value %.1f ms/step %.4f | entry_points %s | knn us %.2f frac %.4f traffic %s" % (d["value"], d["ms_per_step"], {k: r(v) for k, v in d["entry_points"].items() if k != "note"},
      1e3 * d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"].get("traffic")))
print("  kernels", {k: r(v, 4) for k, v in (d.get("kernels") or {}).items() if isinstance(v, float)})
print("  parity", d.get("parity_vs_oracle_m_rad"), d.get("parity_iteration_counts_and_histograms_equal"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
b = d.get("batch64") or {}
print("  batch64 %s by size %s" % (r(b.get("value")), {k: r(v) for k, v in (b.get("registrations_per_s_by_batch_size") or {}).items()}))
for k, v in (d.get("stock") or {}).items():
    if isinstance(v, dict): print("  stock", k, {a: r(b_, 4) for a, b_ in v.items() if a not in ("config", "note")})
o = d.get("open_scene") or {}
print("  open_scene", {a: r(b_, 4) for a, b_ in o.items() if a not in ("workload", "pack_light")})
print("  localization", {k: r(v, 4) for k, v in (d.get("localization") or {}).items() if k != "note"})
PY
}
if [[ $ST == *t* ]]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -60 > $O/pytest_gpu.log; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2 | tee $O/pytest_gpu.txt
fi
if [[ $ST == *b* ]]; then
  T0=$SECONDS; timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json; echo "default bench wall $((SECONDS - T0)) s"; line $O/bench_line.json
  for r in 1 2; do T0=$SECONDS; timeout 900 python bench.py --steps 20 --warmup 5 2>> $O/bench.err | tail -1 > $O/bench_line_steps20_run$r.json; echo "steps20 run $r wall $((SECONDS - T0)) s"; line $O/bench_line_steps20_run$r.json | head -2; done
fi
if [[ $ST == *k* ]]; then
  bash tools/prof_stats.sh $TAG 2>&1 | tail -24 | tee $O/prof_stats.txt
  cp gpurun_out/prof_$TAG/*.csv $O/ 2>/dev/null; cp gpurun_out/prof_$TAG/bench_line.json $O/bench_line_under_rocprofv3.json; cp gpurun_out/prof_$TAG/bench_line_no_speculation.json $O/bench_line_under_rocprofv3_no_speculation.json
fi
if [[ $ST == *m* ]]; then
  bash tools/pmc_traffic.sh $TAG 2>&1 | tail -1 > $O/pmc_traffic.log; mkdir -p $O/pmc; cp gpurun_out/pmc_$TAG/* $O/pmc/ 2>/dev/null; cat $O/pmc_traffic.log | cut -c1-600
  bash tools/pmc_knn.sh 0 2>&1 | tail -18 > $O/pmc/sq_counters_knn_solve.txt; cp gpurun_out/pmc_knn/knn_counters.json $O/pmc/; tail -12 $O/pmc/sq_counters_knn_solve.txt
fi
if [[ $ST == *p* ]]; then
  SOICP_ABLATE=128 python tools/eval_stamps.py 2>&1 | tail -30 | tee $O/phase_stamps_instrumented_build.txt
fi
if [[ $ST == *l* ]]; then
  python tools/localization_rate.py 2>&1 | tail -2 | tee $O/localization_rate.txt
fi
if [[ $ST == *g* ]]; then
  T0=$SECONDS; timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_gpus2.err > $O/bench_gpus2.out
  echo "gpus2 rc=$? wall $((SECONDS - T0)) s stdout lines: $(grep -c . $O/bench_gpus2.out)"; tail -3 $O/bench_gpus2.err
  tail -1 $O/bench_gpus2.out > $O/bench_gpus2.json; line $O/bench_gpus2.json | head -1
fi
