cd $GRAFT_REPO_ROOT
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py 2> gpurun_out/b_final.err | tail -1 > gpurun_out/bench_line_final.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_line_final.json"))
print("value", round(d["value"],1), d["entry_points"], "knn", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "valu", (d["roofline"]["valu_issue"] or {}).get("valu_issue_frac"))
print("batch64", d["batch64"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline_oracle_b"]["value"], d["cpu_baseline_all_cores"]["value"], d["parity_vs_oracle_m_rad"], d["parity_iteration_counts_and_histograms_equal"])
PY
python bench.py --steps 20 --warmup 5 2>> gpurun_out/b_final.err | tail -1 > gpurun_out/bench_line_final20.json; python -c "
import json; d=json.load(open('gpurun_out/bench_line_final20.json')); print('steps20 value', round(d['value'],1), d['entry_points'], d['roofline']['traffic'])"
for m in map queries; do bash tools/two_rank_one_gpu.sh 2 $m 2>&1 | tail -4 | cut -c1-200; cp gpurun_out/ranks2.json gpurun_out/ranks2_$m.json; done
