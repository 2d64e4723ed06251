#!/bin/bash
# one replayed synchronous iteration per kernel under rocprofv3, for an environment switch ON and OFF:  prof_iter_ab.sh VAR [tag]
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R GPU_MAX_HW_QUEUES=2; V=$1; TAG=${2:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for f in 0 1; do
  rm -rf /tmp/p_it; env $V=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_it -- python $R/tools/iter_profile.py 50 "${@:3}" > /dev/null 2>&1
  python $R/tools/summarize_prof.py stats /tmp/p_it 52 > $O/iteration_kernel_stats_${V}_$f.txt
  head -3 $O/iteration_kernel_stats_${V}_$f.txt
done
