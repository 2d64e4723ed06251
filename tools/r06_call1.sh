#!/bin/bash
# round 6, first GPU call: the new parity tests, then the 8-rank (gloo, one shared device) de-risking runs of bench.py and main.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O
export PYTHONPATH=$R; cd $R
timeout 1500 python -m pytest tests/test_drivers_gpu.py -x -q -k "device_side_numpy or two_schedules or mfma_actor_step_keeps or numpy_rng or burn_in" > $O/pytest_new.txt 2>&1
tail -30 $O/pytest_new.txt
(BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-shards > $O/bench_selflaunch_8ranks_gloo_1gpu.json 2> $O/bench_8ranks.err)
tail -c 600 $O/bench_selflaunch_8ranks_gloo_1gpu.json; tail -5 $O/bench_8ranks.err
(ATR_DIST_BACKEND=gloo ATR_SINGLE_DEVICE=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
   main.py --shared-optimizer --split --train-mode -1 --env Track2D-BlockPartialPZR-v0 --num-envs 512 --max-step 50 --test-every 25 --log-every 10 \
   --burn-in 10 --log-dir gpurun_out/r06a/main8_logs/ > $O/main_py_8ranks_gloo_1gpu.txt 2>&1)
tail -15 $O/main_py_8ranks_gloo_1gpu.txt
find $O/main8_logs -name "*.dat" -delete 2>/dev/null
