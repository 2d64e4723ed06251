#!/bin/bash
# rollout / learner CU partitions of the pipelined schedule, ONE partition per process (every masked stream takes a hardware queue
# of its own: partitions tried one after another in one process pile up queues and time-slice each other):
#   cu_split_sweep.sh ENVS SPLIT...      SPLIT = CUs of the rollout stream (the learner's gets the rest); 0 = shared chip (tune_streams)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
N=$1; shift
for s in "$@"; do
  if [ "$s" = "0" ]; then echo "== shared chip (tune_streams picks)"; python tools/shard_sweep.py --schedule pipelined $N 2>&1 | grep shard
  else echo "== rollout $s CUs / learner $((256 - s)) CUs"; ATR_PIPE_CU_SPLIT=$s python tools/shard_sweep.py --schedule pipelined $N 2>&1 | grep shard; fi
done
