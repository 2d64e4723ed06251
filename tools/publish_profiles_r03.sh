#!/bin/bash
# Copies the round-3 evidence from gpurun_out/r03/ (scratch, merged back by gpurun) into profiles/ (tracked), r03_ prefixed.
cd "$(dirname "$0")/.."; O=gpurun_out/r03; P=profiles
cp $O/bench.json $P/r03_bench.json
cp $O/bench_under_rocprof.json $P/r03_bench_under_rocprof.json
cp $O/bench_selflaunch_2ranks_gloo_1gpu.json $P/r03_bench_selflaunch_2ranks_gloo_1gpu.json
cp $O/bench_kernel_stats.txt $P/r03_bench_kernel_stats.txt
cp $O/iteration_kernel_stats.txt $P/r03_iteration_kernel_stats.txt
for n in 512 1024 2048; do cp $O/iter_stats_final_$n.txt $P/r03_iteration_kernel_stats_shard$n.txt; done
cp $O/iter_stats_final_config1.txt $P/r03_iteration_kernel_stats_config1.txt
cp $O/iter_stats_final_config3.txt $P/r03_iteration_kernel_stats_config3.txt
for f in config_sweep shard_sweep pipeline_sweep pipe_timeline corun_kernels_512 two_queue_microbench grid_barrier_microbench gemm_group_bench act_step_bench pair_gemm_bench bptt_bench env_only_other_configs nav_profile \
         learning_check_ram_tracker_synchronous \
         learning_check_ram_tracker learning_check_pzr_dueling learning_check_nav_tracker main_py_logger main_py_scalars_tail; do
  [ -f $O/$f.txt ] && cp $O/$f.txt $P/r03_$f.txt
done
for n in 4096 65536 262144 1048576; do
  cp $O/env_only_$n.txt $P/r03_env_only_$n.txt; cp $O/env_only_kernel_stats_$n.txt $P/r03_env_only_kernel_stats_$n.txt
done
for n in 4096 262144 1048576; do for c in FETCH_SIZE WRITE_SIZE; do cp $O/env_only_pmc_${c}_$n.txt $P/r03_env_only_pmc_${c}_$n.txt; done; done
for c in FETCH_SIZE WRITE_SIZE; do cp $O/act_step_pmc_${c}_4096.txt $P/r03_act_step_pmc_${c}_4096.txt; done
ls $P | grep r03_ | wc -l
python tools/make_pmc_traffic_json.py
