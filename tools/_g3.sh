cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "staged or flags" 2>&1 | tail -15
bash tools/round4_gpu.sh r04c g
# twice more for noise, K=20 only, interleaved
for rep in 1 2; do for at in 0 1 2; do echo "== rep $rep at $at"; SOICP_STAGE_AT=$at timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile-pass --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['host']['c_abi_ms_per_step'], d['host']['fixed_overhead_ms_per_step'])"; done; done
for rep in 1 2; do echo "== resident rep $rep"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile-pass --no-secondary --entry resident 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['host']['c_abi_ms_per_step'], d['host']['fixed_overhead_ms_per_step'])"; done
