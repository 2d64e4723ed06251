# ordered kernel sequence of one replayed iteration at a shard size:  tools/prof_sequence.sh N   (on the GPU box)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_seq
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_seq -- python $R/tools/iter_profile.py 20 Track2D-BlockPartialPZR-v0 $1 tat-maze-lstm reward -1 > /dev/null 2>&1
python $R/tools/iter_sequence.py /tmp/p_seq > $O/iter_sequence_$1.txt
