# kernel trace of the pipelined schedule at a shard size, per-queue summary:  tools/prof_pipe.sh N   (on the GPU box)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r03; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_pipe
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_pipe -- python $R/tools/pipe_probe.py $1 > /tmp/p_pipe.log 2>&1
tail -3 /tmp/p_pipe.log
python $R/tools/pipe_trace.py /tmp/p_pipe > $O/pipe_trace_$1.txt; cat $O/pipe_trace_$1.txt
