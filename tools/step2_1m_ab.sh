# k_step2 at 1 048 576 envs: the product library against one built with round 3's env sources (csrc/track2d_hip.hip + t2d_device.h of
# commit 19a1551, everything else current), three INTERLEAVED passes on ONE box (VERDICT r04 "Next round" 6: is 362 -> 406 us between
# profiles/r03_env_only_kernel_stats_1048576.txt and r04's a regression of the code or the spread between boxes?).
#   tools/step2_1m_ab.sh <r03-env-lib.so> > profiles/r05_step2_1m_ab.txt
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
OLD=$1
for N in 1048576 262144; do for rep in 1 2 3; do for lib in "" "$OLD"; do
  rm -rf /tmp/p_x; T2D_LIB_PATH=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -- python $R/tools/env_only_bench.py --n $N --steps 1000 > /tmp/p_x.log 2>&1
  echo "N=$N pass $rep lib=${lib:-product (round 5 tree)}: k_step2 $(python $R/tools/summarize_prof.py stats /tmp/p_x | grep 'k_step2' | head -1 | awk '{print $(NF-1)}') us per launch"
done; done; done
