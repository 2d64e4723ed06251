#!/bin/bash
# Regenerates the round's evidence under gpurun_out/<tag>/ on the GPU box (copy what you want judged to profiles/).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# PMC passes are separate runs with no trace domains besides the counter collection (pool rule).
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_bench > $O/bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env -- python $R/tools/env_only_bench.py --n 4096 --steps 1000 > $O/env_only_4096.txt 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_env > $O/env_only_kernel_stats.txt
for n in 4096 262144; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$n -- python $R/tools/env_only_bench.py --n $n --steps 200 --warmup 20 > /dev/null 2>&1
  python $R/tools/summarize_prof.py pmc /tmp/p_fetch_$n > $O/env_only_pmc_fetch_$n.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$n -- python $R/tools/env_only_bench.py --n $n --steps 200 --warmup 20 > /dev/null 2>&1
  python $R/tools/summarize_prof.py pmc /tmp/p_write_$n > $O/env_only_pmc_write_$n.txt
done
for n in 4096 65536 262144 1048576; do python $R/tools/env_only_bench.py --n $n --steps 500 --warmup 50; done > $O/env_only_sweep.txt 2>&1
for n in 4096 65536; do python $R/tools/env_only_bench.py --n $n --steps 1000 --warmup 100 --fused; done > $O/env_only_fused.txt 2>&1
for e in Track2D-BlockPartialRam-v0 Track2D-MazePartialNav-v0 Track2D-BlockPartialAdv-v0; do python $R/tools/env_only_bench.py --n 8192 --env $e --steps 300 --warmup 30; done > $O/env_only_other_configs.txt 2>&1
python $R/tools/stem_bench.py > $O/stem_bench.txt 2>&1
python $R/tools/learning_check.py --iters 1000 > $O/learning_check_ram_tracker.txt 2>&1
python $R/tools/learning_check.py --iters 600 --env Track2D-BlockPartialPZR-v0 --network tat-maze-lstm --train-mode -1 > $O/learning_check_pzr_dueling.txt 2>&1
ls -la $O
cat $O/bench.json | cut -c1-2500
cat $O/env_only_pmc_*.txt $O/env_only_sweep.txt $O/env_only_fused.txt $O/env_only_other_configs.txt
