R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_under_rocprof.json 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_bench > $O/bench_kernel_stats.txt
head -12 $O/bench_kernel_stats.txt
for n in 4096 65536; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env$n -- python $R/tools/env_only_bench.py --n $n --steps 1000 > $O/env_only_$n.txt 2> /dev/null
python $R/tools/summarize_prof.py stats /tmp/p_env$n > $O/env_only_kernel_stats_$n.txt; head -6 $O/env_only_kernel_stats_$n.txt
done
