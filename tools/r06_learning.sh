#!/bin/bash
# round 6: configs[3] learning table at the SHIPPED start (main.py --burn-in 150; the pipelined schedule also runs tune_streams first),
# 16 seeds x 3000 iterations per schedule, with the first 200 iterations of seeds 5 and 7 traced under both schedules
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R; O=$R/gpurun_out/r06c; mkdir -p $O
S="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16"
(echo "# (1) the shipped defaults: --burn-in 150 for both schedules; pipelined after tune_streams() (main.py)"
 python tools/learning_seeds.py --burn-in 150 --seeds $S --schedules synchronous pipelined --trace-seeds 5 7 2>&1 | grep -v amdgpu.ids
 echo; echo "# (2) the same start for the one-stream form of the pipelined dataflow (no tune_streams: 150 untrained iterations, as synchronous)"
 python tools/learning_seeds.py --burn-in 150 --seeds $S --schedules pipelined-serial 2>&1 | grep -v amdgpu.ids
 echo; echo "# (3) synchronous with the untrained prelude the shipped pipelined start has in all (tune_streams ~ 280 + 150)"
 python tools/learning_seeds.py --burn-in 430 --seeds $S --schedules synchronous 2>&1 | grep -v amdgpu.ids) > $O/learning_seeds_nav_mode0_burnin150.txt
tail -12 $O/learning_seeds_nav_mode0_burnin150.txt
