R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 Track2D-BlockPartialRam-v0 1024 maze-lstm none 0 > /dev/null 2>&1
python $R/tools/summarize_prof.py stats /tmp/p_it > $O/iteration_kernel_stats_cfg1.txt
head -45 $O/iteration_kernel_stats_cfg1.txt | cut -c1-90,113-160
