#!/bin/bash
# Copies the round-4 evidence from gpurun_out/r04/ (scratch, merged back by gpurun) into profiles/ (tracked), r04_ prefixed.
cd "$(dirname "$0")/.."; O=gpurun_out/r04; P=profiles
for f in bench.json bench_driver_flags.json bench_under_rocprof.json bench_selflaunch_2ranks_gloo_1gpu.json; do [ -s $O/$f ] && cp $O/$f $P/r04_$f; done
for f in bench_kernel_stats iteration_kernel_stats iteration_kernel_stats_shard2048 iteration_kernel_stats_shard1024 iteration_kernel_stats_shard512 \
         iteration_kernel_stats_config1 iteration_kernel_stats_config3 config_sweep lt_gemm_bench act_step_bench gemm_group_bench gemm_tn_timeline multirank_1gpu \
         nav_env_only_1024 nav_env_only_8192 nav_kernel_stats_1024 nav_kernel_stats_8192 generator_nav_timeline \
         learning_check_ram_tracker learning_check_pzr_dueling learning_check_nav_tracker main_py_logger main_py_scalars_tail \
         main_py_test_scalars_tail; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $P/r04_$f.txt
done
for n in 4096 65536 262144 1048576; do
  [ -s $O/env_only_$n.txt ] && cp $O/env_only_$n.txt $P/r04_env_only_$n.txt; [ -s $O/env_only_kernel_stats_$n.txt ] && cp $O/env_only_kernel_stats_$n.txt $P/r04_env_only_kernel_stats_$n.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  cp $O/env_only_pmc_${c}_4096.txt $P/r04_env_only_pmc_${c}_4096.txt; cp $O/act_step_pmc_${c}_4096.txt $P/r04_act_step_pmc_${c}_4096.txt
done
ls $P | grep r04_ | wc -l
python tools/make_pmc_traffic_json.py r04
