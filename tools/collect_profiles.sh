#!/bin/bash
# Regenerates the round's evidence under gpurun_out/<tag>/ on the GPU box (copy what you want judged to profiles/).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh r02'
# PMC passes are separate runs with no trace domains besides the counter collection (pool rule).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_bench > $O/bench_kernel_stats.txt
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 900 python $R/bench.py --gpus 2 --steps 100 --warmup 20 > $O/bench_selflaunch_2ranks_gloo_1gpu.json 2> $O/bench_selflaunch_2ranks_gloo_1gpu.err
for n in 4096 65536 262144; do
  rm -rf /tmp/p_env; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env -- python $R/tools/env_only_bench.py --n $n --steps 1000 > $O/env_only_$n.txt 2> /dev/null
  python $R/tools/summarize_prof.py stats /tmp/p_env > $O/env_only_kernel_stats_$n.txt
done
bash $R/tools/pmc_traffic.sh $TAG 4096 65536 262144 > /dev/null 2>&1
for n in 4096 65536 262144 1048576; do timeout 300 python $R/tools/env_only_bench.py --n $n --steps 500 --warmup 50; done > $O/env_only_sweep.txt 2>&1
for n in 4096 65536; do timeout 300 python $R/tools/env_only_bench.py --n $n --steps 1000 --warmup 100 --fused; done > $O/env_only_fused.txt 2>&1
for e in Track2D-BlockPartialRam-v0 Track2D-MazePartialNav-v0 Track2D-BlockPartialAdv-v0; do timeout 300 python $R/tools/env_only_bench.py --n 8192 --env $e --steps 300 --warmup 30; done > $O/env_only_other_configs.txt 2>&1
timeout 300 python $R/tools/hbm_ceiling.py > $O/hbm_ceiling.txt 2>&1
timeout 300 python $R/tools/stem_bench.py > $O/stem_bench.txt 2>&1
timeout 300 python $R/tools/stem_u8_ab.py > $O/stem_u8_ab.txt 2>&1
timeout 300 python $R/tools/actor_step_bench.py > $O/actor_step_bench.txt 2>&1
bash $R/tools/prof_nav.sh > $O/nav_profile.txt 2>&1
timeout 600 python $R/tools/config_sweep.py > $O/config_sweep.txt 2>&1
if [ -f $R/scratch_exp/libexp6.so ]; then
  T2D_LIB_PATH=$R/scratch_exp/libexp6.so timeout 120 python $R/tools/timeline_probe.py 4096 > $O/step_kernel_timeline_4096.txt 2>&1
  (bash $R/tools/exp_variants.sh 5 7 8; N=65536 bash $R/tools/exp_variants.sh 5 7 8) > $O/step_kernel_latency_probes.txt 2>&1
fi
timeout 300 bash $R/tools/prof_iter.sh
ls -la $O
