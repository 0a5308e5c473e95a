cd $GRAFT_REPO_ROOT
echo "== PMC, packing on"; bash tools/pmc_knn.sh 0 2>&1 | grep -E "knn_plane|solve_kernel" | grep -E "INSTS_VALU|INSTS_SALU|ACTIVE_INST_VALU|WAVE_CYCLES|SQ_WAVES|INSTS_LDS"
cp gpurun_out/pmc_knn/knn_counters.json gpurun_out/pmc_knn/knn_counters_pack.json; cp gpurun_out/pmc_knn/ablate_0.txt gpurun_out/pmc_knn/sq_counters_pack.txt
echo "== PMC, packing off (SOICP_KNN_PACK=0)"; SOICP_KNN_PACK=0 bash tools/pmc_knn.sh 0 2>&1 | grep -E "knn_plane" | grep -E "INSTS_VALU|INSTS_SALU|ACTIVE_INST_VALU|WAVE_CYCLES|SQ_WAVES|INSTS_LDS"
cp gpurun_out/pmc_knn/ablate_0.txt gpurun_out/pmc_knn/sq_counters_nopack.txt; cp gpurun_out/pmc_knn/knn_counters_pack.json gpurun_out/pmc_knn/knn_counters.json
