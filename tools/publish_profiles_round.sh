#!/bin/bash
# Copies a round's evidence from gpurun_out/<tag>/ (scratch, merged back by gpurun) into profiles/ (tracked), <tag>_ prefixed:
#   bash tools/publish_profiles_round.sh r06
TAG=${1:-r06}; cd "$(dirname "$0")/.."; O=gpurun_out/$TAG; P=profiles
for f in bench.json bench_driver_flags.json bench_under_rocprof.json bench_selflaunch_2ranks_gloo_1gpu.json bench_selflaunch_8ranks_gloo_1gpu.json; do [ -s $O/$f ] && cp $O/$f $P/${TAG}_$f; done
for f in bench_kernel_stats iteration_kernel_stats iteration_kernel_stats_shard2048 iteration_kernel_stats_shard1024 iteration_kernel_stats_shard512 \
         iteration_kernel_stats_config1 iteration_kernel_stats_config3 config_sweep lt_gemm_bench act_step_bench gemm_group_bench gemm_tn_timeline multirank_1gpu \
         nav_env_only_1024 nav_env_only_8192 nav_kernel_stats_1024 nav_kernel_stats_8192 generator_nav_timeline \
         learning_check_ram_tracker learning_check_pzr_dueling learning_check_nav_tracker main_py_logger main_py_scalars_tail \
         main_py_test_scalars_tail coop_step_timeline_512 coop_step_timeline_1024 shard_sweep shard_sweep_coop_step nav_env_only_1024_pregrow nav_kernel_stats_1024_pregrow \
         iteration_kernel_stats_config3_pipelined xcd_barrier_microbench stem_bench stem_rollout_bench stem_timelines mfma_order_microbench pytest_gpu \
         gate_cell_bench cu_split_sweep_512 main_py_8ranks_gloo_1gpu iteration_kernel_stats_ATR_GATE_CELL_0 iteration_kernel_stats_ATR_GATE_CELL_1 \
         iteration_kernel_stats_ATR_FOLD_EMBEDDING_0 iteration_kernel_stats_ATR_FOLD_EMBEDDING_1; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $P/${TAG}_$f.txt
done
for n in 4096 65536 262144 1048576; do
  [ -s $O/env_only_$n.txt ] && cp $O/env_only_$n.txt $P/${TAG}_env_only_$n.txt; [ -s $O/env_only_kernel_stats_$n.txt ] && cp $O/env_only_kernel_stats_$n.txt $P/${TAG}_env_only_kernel_stats_$n.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  cp $O/env_only_pmc_${c}_4096.txt $P/${TAG}_env_only_pmc_${c}_4096.txt; cp $O/act_step_pmc_${c}_4096.txt $P/${TAG}_act_step_pmc_${c}_4096.txt
done
ls $P | grep ${TAG}_ | wc -l
python tools/make_pmc_traffic_json.py $TAG
