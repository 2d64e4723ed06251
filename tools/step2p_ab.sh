# k_step2 (one wave per env pair) vs k_step2p (persistent, pipelined) at large batch sizes: parity test, then kernel times
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persistent" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
for n in 65536 262144 1048576; do for min in 999999999 1; do for bpc in ${BPCS:-0}; do
  rm -rf /tmp/p_s2; T2D_STEP2P_MIN=$min T2D_STEP2P_BLOCKS_PER_CU=$bpc timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s2 -- python $R/tools/env_only_bench.py --n $n --steps 200 --warmup 40 > /dev/null 2>&1
  echo "N=$n T2D_STEP2P_MIN=$min blocks/CU=$bpc: $(python $R/tools/summarize_prof.py stats /tmp/p_s2 | grep 'k_step2' | cut -c1-30,105-160)"
done; done; done
