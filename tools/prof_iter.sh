R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 > /dev/null 2>&1
python $R/tools/summarize_prof.py stats /tmp/p_it > $O/iteration_kernel_stats.txt
