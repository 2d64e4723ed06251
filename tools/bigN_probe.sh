# Is k_step2 at large N bound by bytes or by latency x occupancy? Full kernel vs the same kernel without observation stores.
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd $R
for n in 262144 1048576; do
  timeout 120 python tools/env_only_bench.py --n $n --steps 300 --warmup 50 2>/dev/null | tail -1
  timeout 120 python tools/env_only_bench.py --n $n --steps 300 --warmup 50 --no-obs 2>/dev/null | tail -1
done
