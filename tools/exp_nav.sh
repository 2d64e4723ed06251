R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
for lib in "" $R/scratch_exp/libnomaze.so; do
rm -rf /tmp/p_nav; T2D_LIB_PATH=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nav -- python $R/tools/env_only_bench.py --env Track2D-MazePartialNav-v0 --n 1024 --steps 600 --warmup 100 > /dev/null 2>&1
echo "lib=${lib:-product}"; python $R/tools/summarize_prof.py stats /tmp/p_nav | grep "k_gen\|k_env<0" | cut -c1-50,105-160
done
