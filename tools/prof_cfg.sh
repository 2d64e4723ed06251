# per-iteration kernel table of any configuration:  tools/prof_cfg.sh TAG ENV N NETWORK AUX MODE   (on the GPU box)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 $2 $3 $4 $5 $6 > /dev/null 2>&1
python $R/tools/summarize_prof.py stats /tmp/p_it 52 > $O/iter_stats_$1.txt
