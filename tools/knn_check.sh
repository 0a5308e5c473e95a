python tools/eval_stamps.py 2>&1 | grep -A3 "knn sweep"; python bench.py --no-cpu-baseline --steps 48 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity_vs_oracle_m_rad'] if 'parity_vs_oracle_m_rad' in d else '', d['executed'])"
