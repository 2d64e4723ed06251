#!/bin/bash
# Round 3 evidence, regenerated under gpurun_out/r03/ on the GPU box (copy what is to be judged to profiles/):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles_r03.sh'
# PMC passes are separate runs with no trace domains besides the counter collection (pool rule).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
# --- the bench line, and the same command under the profiler
timeout 1200 python $R/bench.py > $O/bench.json 2> $O/bench.err
rm -rf /tmp/p_bench; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --no-shards > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_bench > $O/bench_kernel_stats.txt
# --- the launcher + both schedules with a real process group: 2 ranks sharing the one GPU over gloo (not a scaling number)
(cd $R && BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_selflaunch_2ranks_gloo_1gpu.json 2> /dev/null)
# --- one replayed iteration per kernel: the headline batch and every shard size, and the other BASELINE configurations
for n in 4096 2048 1024 512; do bash $R/tools/prof_shard.sh $n final; done
cp $O/iter_stats_final_4096.txt $O/iteration_kernel_stats.txt
bash $R/tools/prof_cfg.sh final_config1 Track2D-BlockPartialRam-v0 1024 maze-lstm none 0
bash $R/tools/prof_cfg.sh final_config3 Track2D-MazePartialNav-v0 1024 maze-lstm none 0
timeout 600 python $R/tools/config_sweep.py > $O/config_sweep.txt 2>&1
timeout 600 python $R/tools/shard_sweep.py > $O/shard_sweep.txt 2>&1
SWEEP_ORDER=sp timeout 600 python $R/tools/pipeline_sweep.py > $O/pipeline_sweep.txt 2>&1
# --- the pipelined schedule: who waits for whom (HIP events), what slows the rollout chain next to the learner
for n in 512 4096; do timeout 300 python $R/tools/pipe_timeline.py $n; done > $O/pipe_timeline.txt 2>&1
timeout 300 python $R/tools/corun_kernels.py 512 > $O/corun_kernels_512.txt 2>&1
[ -x $R/scratch_exp/two_queue ] && TWOQ_MEM=1 timeout 120 $R/scratch_exp/two_queue > $O/two_queue_microbench.txt 2>&1
[ -x $R/scratch_exp/grid_barrier ] && timeout 120 $R/scratch_exp/grid_barrier > $O/grid_barrier_microbench.txt 2>&1
# --- the round's kernels alone
timeout 300 python $R/tools/act_step_bench.py > $O/act_step_bench.txt 2>&1
timeout 300 python $R/tools/pair_gemm_bench.py > $O/pair_gemm_bench.txt 2>&1
timeout 300 python $R/tools/bptt_bench.py > $O/bptt_bench.txt 2>&1
timeout 300 python $R/tools/gemm_group_bench.py > $O/gemm_group_bench.txt 2>&1
# --- learning checks (tracker vs Ram, the PZR duel, tracker vs Nav in mazes) under the default (pipelined) schedule, and a main.py run
timeout 600 python $R/tools/learning_check.py --iters 1500 > $O/learning_check_ram_tracker.txt 2>&1
timeout 600 python $R/tools/learning_check.py --iters 1500 --schedule synchronous > $O/learning_check_ram_tracker_synchronous.txt 2>&1
timeout 600 python $R/tools/learning_check.py --env Track2D-BlockPartialPZR-v0 --network tat-maze-lstm --train-mode -1 --iters 1500 > $O/learning_check_pzr_dueling.txt 2>&1
timeout 600 python $R/tools/learning_check.py --env Track2D-MazePartialNav-v0 --num-envs 1024 --iters 1500 > $O/learning_check_nav_tracker.txt 2>&1
rm -rf $O/main_logs; (cd $R && timeout 900 python main.py --shared-optimizer --split --train-mode -1 --env Track2D-BlockPartialPZR-v0 --num-envs 4096 \
    --max-step 2000 --test-every 200 --log-dir gpurun_out/r03/main_logs/ > $O/main_py_run.txt 2>&1)
cp $(ls -d $O/main_logs/*/*/ | head -1)logger $O/main_py_logger.txt 2>/dev/null
tail -40 $(ls $O/main_logs/*/*/Agent:0/scalars.jsonl | head -1) > $O/main_py_scalars_tail.txt 2>/dev/null
find $O/main_logs -name "*.dat" -delete
# --- env-only: step kernel alone at every size (rocprofv3 stats), sweep, the other env ids, Nav split
for n in 4096 65536 262144 1048576; do
  rm -rf /tmp/p_env; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env -- python $R/tools/env_only_bench.py --n $n --steps 600 --warmup 100 > $O/env_only_$n.txt 2> /dev/null
  python $R/tools/summarize_prof.py stats /tmp/p_env > $O/env_only_kernel_stats_$n.txt
done
for e in Track2D-BlockPartialRam-v0 Track2D-MazePartialNav-v0 Track2D-BlockPartialAdv-v0; do timeout 300 python $R/tools/env_only_bench.py --n 8192 --env $e --steps 300 --warmup 30; done > $O/env_only_other_configs.txt 2>&1
bash $R/tools/prof_nav.sh > $O/nav_profile.txt 2>&1
# --- HBM traffic (PMC): FETCH_SIZE and WRITE_SIZE in separate passes, env-only step kernel at three sizes ...
bash $R/tools/pmc_traffic.sh r03 4096 262144 1048576 > /dev/null 2>&1
# ... and the fused end-of-step kernel of the timed region (k_act_step) + the byte-observation step kernel, from act_step_bench
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_pmc; ACT_BENCH_MODE=sep timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_pmc -- python $R/tools/act_step_bench.py 4096 > /dev/null 2>&1
  python $R/tools/summarize_prof.py pmc /tmp/p_pmc k_act_step > $O/act_step_pmc_${c}_4096.txt
done
ls -la $O
