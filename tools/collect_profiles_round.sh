#!/bin/bash
# A round's evidence, regenerated under gpurun_out/<tag>/ on the GPU box (tools/publish_profiles_round.sh <tag> copies what is to
# be judged to profiles/):   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/collect_profiles_round.sh r06'
# PMC passes are separate runs with no trace domains besides the counter collection (pool rule).
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export PYTHONPATH=$R GPU_MAX_HW_QUEUES=2; cd /tmp; export TMPDIR=/tmp
# --- the bench line (driver flags and defaults), and the same command under the profiler
timeout 1200 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2> /dev/null
rm -rf /tmp/p_bench; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --no-shards > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_bench > $O/bench_kernel_stats.txt
(cd $R && BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline --no-shards > $O/bench_selflaunch_2ranks_gloo_1gpu.json 2> /dev/null)
# --- one replayed iteration per kernel: the headline batch and every shard size, and the other BASELINE configurations
prof_iter() {   # name, then iter_profile.py arguments
  rm -rf /tmp/p_it; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 "${@:2}" > /dev/null 2>&1
  python $R/tools/summarize_prof.py stats /tmp/p_it 52 > $O/$1.txt
}
prof_iter iteration_kernel_stats
for n in 2048 1024 512; do prof_iter iteration_kernel_stats_shard$n Track2D-BlockPartialPZR-v0 $n tat-maze-lstm reward -1; done
prof_iter iteration_kernel_stats_config1 Track2D-BlockPartialRam-v0 1024 maze-lstm none 0
prof_iter iteration_kernel_stats_config3 Track2D-MazePartialNav-v0 1024 maze-lstm none 0
(cd $R && timeout 600 python tools/config_sweep.py > $O/config_sweep.txt 2>&1)
# --- the round's pieces alone
(cd $R && timeout 300 python tools/lt_gemm_bench.py > $O/lt_gemm_bench.txt 2>&1)
# --- round 5: the conv stems across launch sizes (16 frames per workgroup pass from 16384 frames up), the rollout-shaped launches with
# either forward kernel, and the phase timelines of the 16-frame kernels (probe build)
(cd $R && timeout 300 python tools/stem_bench.py 2>&1 | grep -v amdgpu.ids > $O/stem_bench.txt)
(cd $R && (echo "== wave per frame (k_stem_fwd, k_stem_bwd: ATR_STEM_FWD16_MIN / ATR_STEM_BWD16_MIN out of reach)"; ATR_STEM_FWD16_MIN=100000000 ATR_STEM_BWD16_MIN=100000000 timeout 300 python tools/stem_bench.py 8192 81920 163840 2>&1 | grep -v amdgpu.ids) >> $O/stem_bench.txt)
(cd $R && (echo "== rollout-shaped launches, the default rule (use_fwd16 in csrc/stem_hip.hip: 16 frames per pass from 3072 frames up where the passes load the CUs evenly)"; timeout 300 python tools/stem_rollout_bench.py 2>&1 | grep -v amdgpu.ids; echo "== wave per frame everywhere below 16384 frames (ATR_STEM_FWD16_MIN=16384)"; ATR_STEM_FWD16_MIN=16384 timeout 300 python tools/stem_rollout_bench.py 2>&1 | grep -v amdgpu.ids; echo "== 16 frames per pass forced (ATR_STEM_FWD16_MIN=2048)"; ATR_STEM_FWD16_MIN=2048 timeout 300 python tools/stem_rollout_bench.py 2>&1 | grep -v amdgpu.ids) > $O/stem_rollout_bench.txt)
(cd $R/active_tracking_rl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -ldl -DSTEM_PROBE \
    -o $R/scratch_exp/libstemprobe.so track2d_hip.hip stem_hip.hip policy_hip.hip lstm_hip.hip heads_hip.hip gemm_tn_hip.hip actor_step_hip.hip pair_gemm_hip.hip bptt_hip.hip driver_hip.hip gate_cell_hip.hip np_mode.cpp lt_gemm.cpp > /dev/null 2>&1)
(cd $R && (ATR_STEM_FWD16_MIN=2048 T2D_LIB_PATH=$R/scratch_exp/libstemprobe.so timeout 300 python tools/stem_timeline_probe.py 4096 2>&1 | grep -v amdgpu.ids; echo; echo "== the headline rollout launch (4096 + 8192 frames), per workgroup"; ATR_STEM_FWD16_MIN=2048 T2D_LIB_PATH=$R/scratch_exp/libstemprobe.so timeout 300 python tools/stem_tat_timeline.py 4096 2>&1 | grep -v amdgpu.ids; echo; T2D_LIB_PATH=$R/scratch_exp/libstemprobe.so timeout 300 python tools/stem_bwd_timeline_probe.py 81920 2>&1 | grep -v amdgpu.ids) > $O/stem_timelines.txt)
# --- round 6: is the f32 MFMA's k-sum the ascending fmaf chain? (what the stems' bit-identity between kernels rests on)
(cd $R/tools/microbench && hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_order.hip -o $R/scratch_exp/mfma_order > /dev/null 2>&1; timeout 60 $R/scratch_exp/mfma_order > $O/mfma_order_microbench.txt 2>&1)
(cd $R && timeout 300 python tools/act_step_bench.py > $O/act_step_bench.txt 2>&1)
(cd $R && ACT_BENCH_MODE=one timeout 300 python tools/act_step_bench.py 512 1024 2048 4096 >> $O/act_step_bench.txt 2>&1)
(cd $R && ACT_BENCH_MODE=pre timeout 300 python tools/act_step_bench.py 512 1024 2048 4096 >> $O/act_step_bench.txt 2>&1)
# --- the learner's grouped weight-gradient GEMM alone, and its per-workgroup timeline (probe build -DATR_TN_PROBE=1, compiled here)
(cd $R && timeout 300 python tools/gemm_group_bench.py 512 1024 2048 4096 > $O/gemm_group_bench.txt 2>&1)
mkdir -p $R/scratch_exp; (cd $R/active_tracking_rl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -ldl -Wno-unused-result -DATR_TN_PROBE=1 \
    -o $R/scratch_exp/libtnprobe.so track2d_hip.hip stem_hip.hip policy_hip.hip lstm_hip.hip heads_hip.hip gemm_tn_hip.hip actor_step_hip.hip pair_gemm_hip.hip bptt_hip.hip driver_hip.hip gate_cell_hip.hip np_mode.cpp lt_gemm.cpp > /dev/null 2>&1)
[ -f $R/scratch_exp/libtnprobe.so ] && (cd $R && T2D_LIB_PATH=scratch_exp/libtnprobe.so timeout 300 python tools/gemm_tn_timeline.py 4096 1536 > $O/gemm_tn_timeline.txt 2>&1)
# --- multi-rank settings under a 1-rank RCCL group
(cd $R && bash tools/multirank_probe.sh > /dev/null 2>&1; cp gpurun_out/r04_multirank_1gpu.txt $O/multirank_1gpu.txt 2>/dev/null)
# --- Nav / Maze: generator pass (kernel stats at 1024 / 8192 random-policy envs), its timeline (probe build, if present)
for n in 1024 8192; do
  rm -rf /tmp/p_nav; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nav -- python $R/tools/env_only_bench.py --env Track2D-MazePartialNav-v0 --n $n --steps 600 --warmup 100 > $O/nav_env_only_$n.txt 2>/dev/null
  python $R/tools/summarize_prof.py stats /tmp/p_nav > $O/nav_kernel_stats_$n.txt
done
# (round 5) the same loop with the Maze maps grown ahead of the pass (t2d_pregrow, forked behind every pass), and the pipelined
# iteration of configs[3], where k_pregrow runs on the learner's stream
rm -rf /tmp/p_nav; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nav -- python $R/tools/env_only_bench.py --env Track2D-MazePartialNav-v0 --n 1024 --steps 600 --warmup 100 --pregrow > $O/nav_env_only_1024_pregrow.txt 2>/dev/null
python $R/tools/summarize_prof.py stats /tmp/p_nav > $O/nav_kernel_stats_1024_pregrow.txt
rm -rf /tmp/p_it; ITER_PROFILE_SCHEDULE=pipelined timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 200 Track2D-MazePartialNav-v0 1024 maze-lstm none 0 > /dev/null 2>&1
python $R/tools/summarize_prof.py stats /tmp/p_it 200 > $O/iteration_kernel_stats_config3_pipelined.txt
# (the probe build of the library, -DT2D_EXP=9, compiled here from the sources of this very tree)
mkdir -p $R/scratch_exp; (cd $R/active_tracking_rl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -ldl -Wno-unused-result -DT2D_EXP=9 \
    -o $R/scratch_exp/libexp9.so track2d_hip.hip stem_hip.hip policy_hip.hip lstm_hip.hip heads_hip.hip gemm_tn_hip.hip actor_step_hip.hip pair_gemm_hip.hip bptt_hip.hip driver_hip.hip gate_cell_hip.hip np_mode.cpp lt_gemm.cpp > /dev/null 2>&1)
[ -f $R/scratch_exp/libexp9.so ] && (cd $R && T2D_LIB_PATH=scratch_exp/libexp9.so timeout 300 python tools/gen_nav_timeline_probe.py > $O/generator_nav_timeline.txt 2>&1)
# --- env-only: the stand-alone step kernel at every size (rocprofv3 stats)
for n in 4096 65536 262144 1048576; do
  rm -rf /tmp/p_env; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env -- python $R/tools/env_only_bench.py --n $n --steps 600 --warmup 100 > $O/env_only_$n.txt 2> /dev/null
  python $R/tools/summarize_prof.py stats /tmp/p_env > $O/env_only_kernel_stats_$n.txt
done
# --- HBM traffic (PMC): FETCH_SIZE and WRITE_SIZE in separate passes — the stand-alone step kernel at 4096 envs, and the fused
# end-of-step kernel in the form the timed region runs it (one gate tensor, bias, masked hidden rows; since round 5 without the
# activated-gates store: ACT_BENCH_MODE=pre)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_pmc; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_pmc -- python $R/tools/env_only_bench.py --n 4096 --steps 300 --warmup 50 > /dev/null 2>&1
  python $R/tools/summarize_prof.py pmc /tmp/p_pmc k_step2 > $O/env_only_pmc_${c}_4096.txt
  rm -rf /tmp/p_pmc; (cd $R && ACT_BENCH_MODE=pre timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_pmc -- python tools/act_step_bench.py 4096 > /dev/null 2>&1)
  python $R/tools/summarize_prof.py pmc /tmp/p_pmc k_act_step > $O/act_step_pmc_${c}_4096.txt
done
# --- shard sizes (the coop-step timelines of round 5 are not regenerated: that kernel is experimental and unchanged)
(cd $R && timeout 600 python tools/shard_sweep.py 512 1024 2048 4096 > $O/shard_sweep.txt 2>&1)
# --- round 6: the gate product with the cell as its epilogue against the library product (timeline per workgroup), the same iteration
# with it on / off and with the embedding fold on / off, the CU-partition curve one partition per process, the 8-rank runs on one GPU
(cd $R && timeout 600 python tools/gate_cell_bench.py 4096 2048 1024 --timeline 2>&1 | grep -v amdgpu.ids > $O/gate_cell_bench.txt)
bash $R/tools/prof_iter_ab.sh ATR_GATE_CELL $TAG > /dev/null 2>&1
bash $R/tools/prof_iter_ab.sh ATR_FOLD_EMBEDDING $TAG > /dev/null 2>&1
(cd $R && timeout 900 bash tools/cu_split_sweep.sh 512 0 64 96 112 128 144 160 192 > $O/cu_split_sweep_512.txt 2>&1)
(cd $R && BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-shards 2> /dev/null | grep "^{" > $O/bench_selflaunch_8ranks_gloo_1gpu.json)
(cd $R && ATR_DIST_BACKEND=gloo ATR_SINGLE_DEVICE=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
   main.py --shared-optimizer --split --train-mode -1 --env Track2D-BlockPartialPZR-v0 --num-envs 512 --max-step 50 --test-every 25 --log-every 10 \
   --burn-in 10 --log-dir gpurun_out/$TAG/main8_logs/ 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|^20.*: [a-z_]*:" > $O/main_py_8ranks_gloo_1gpu.txt)
rm -rf $O/main8_logs
# --- learning checks under the default (pipelined) schedule, and a main.py run with the evaluator's scalars
(cd $R && timeout 600 python tools/learning_check.py --iters 1500 > $O/learning_check_ram_tracker.txt 2>&1)
(cd $R && timeout 600 python tools/learning_check.py --env Track2D-BlockPartialPZR-v0 --network tat-maze-lstm --train-mode -1 --iters 1500 > $O/learning_check_pzr_dueling.txt 2>&1)
(cd $R && timeout 600 python tools/learning_check.py --env Track2D-MazePartialNav-v0 --num-envs 1024 --iters 1500 > $O/learning_check_nav_tracker.txt 2>&1)
rm -rf $O/main_logs; (cd $R && timeout 900 python main.py --shared-optimizer --split --train-mode -1 --env Track2D-BlockPartialPZR-v0 --num-envs 4096 \
    --max-step 1000 --test-every 250 --log-dir gpurun_out/$TAG/main_logs/ > $O/main_py_run.txt 2>&1)
cp $(ls -d $O/main_logs/*/*/ | head -1)logger $O/main_py_logger.txt 2>/dev/null
tail -30 $(ls $O/main_logs/*/*/Agent:0/scalars.jsonl | head -1) > $O/main_py_scalars_tail.txt 2>/dev/null
tail -12 $(ls $O/main_logs/*/*/Test/scalars.jsonl | head -1) > $O/main_py_test_scalars_tail.txt 2>/dev/null
find $O/main_logs -name "*.dat" -delete
(cd $R && timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1)
ls -la $O
