R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for n in 1024 8192; do
rm -rf /tmp/p_nav; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nav -- python $R/tools/env_only_bench.py --env Track2D-MazePartialNav-v0 --n $n --steps 600 --warmup 100 > $O/nav_env_only_$n.txt 2>/dev/null
python $R/tools/summarize_prof.py stats /tmp/p_nav > $O/nav_kernel_stats_$n.txt
cat $O/nav_env_only_$n.txt | tail -1; head -8 $O/nav_kernel_stats_$n.txt | cut -c1-60,105-160
done
