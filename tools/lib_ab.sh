# A/B of the step kernel's average duration between the product library and experiment builds: tools/lib_ab.sh <N> <lib.so> [...]
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
N=$1; shift
for lib in "" "$@"; do for rep in 1 2 3; do
  rm -rf /tmp/p_x; T2D_LIB_PATH=$lib timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -- python $R/tools/env_only_bench.py --n $N --steps 2000 > /dev/null 2>&1
  echo "N=$N lib=${lib:-product}: $(python $R/tools/summarize_prof.py stats /tmp/p_x | grep 'k_step2' | head -1 | awk '{print $(NF-1)}') us"
done; done
