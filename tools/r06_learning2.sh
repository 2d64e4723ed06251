#!/bin/bash
# round 6: is the synchronous schedule's early policy collapse on configs[3] a matter of step size / entropy weight? (16 seeds each)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out/r06c; mkdir -p $O
S="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16"
(echo "# (4) synchronous, --burn-in 150, learning rate 3e-4 instead of 1e-3"
 python tools/learning_seeds.py --burn-in 150 --seeds $S --schedules synchronous --lr 3e-4 2>&1 | grep -v amdgpu.ids
 echo; echo "# (5) synchronous, --burn-in 150, tracker entropy weight 0.05 instead of 0.01"
 python tools/learning_seeds.py --burn-in 150 --seeds $S --schedules synchronous --entropy 0.05 2>&1 | grep -v amdgpu.ids
 echo; echo "# (6) pipelined (tune_streams + burn-in 150), learning rate 3e-4"
 python tools/learning_seeds.py --burn-in 150 --seeds $S --schedules pipelined --lr 3e-4 2>&1 | grep -v amdgpu.ids) > $O/learning_seeds_nav_mode0_remedies.txt
grep "^# " $O/learning_seeds_nav_mode0_remedies.txt
