# per-iteration kernel table of the headline workload at a shard size:  tools/prof_shard.sh N TAG   (on the GPU box)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 Track2D-BlockPartialPZR-v0 $1 tat-maze-lstm reward -1 > /dev/null 2>&1
python $R/tools/summarize_prof.py stats /tmp/p_it 52 > $O/iter_stats_$2_$1.txt
