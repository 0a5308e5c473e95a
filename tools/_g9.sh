cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -3
for round in 1 2; do for v in nopack pack; do
  if [ $v = packoff ]; then cp superodom_amd/lib/libsoicp_pack.so superodom_amd/lib/libsoicp.so; export SOICP_KNN_PACK=0; else unset SOICP_KNN_PACK; cp superodom_amd/lib/libsoicp_$v.so superodom_amd/lib/libsoicp.so; fi
  echo "== $v (round $round)"
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
print('value %.1f resident %.1f knn us %.2f | per reg knn %.4f solve %.4f bin %.4f | batch64 %.1f' % (d['value'], d['entry_points']['resident'], 1e3*d['roofline']['avg_launch_ms'], k['knn_ms_per_registration'], k['solve_ms_per_registration'], k['binning_ms_per_registration'], d['batch64']['value']))"
done; done
unset SOICP_KNN_PACK; cp superodom_amd/lib/libsoicp_pack.so superodom_amd/lib/libsoicp.so
SOICP_ABLATE=128 python tools/eval_stamps.py --reps 4 2>&1 | grep -E "knn sweep|per chunk|life|slowest chunk"
