# kernel-duration attribution: rocprofv3 average of the step kernel for experiment builds of the library (T2D_EXP)
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
N=${N:-4096}
run() { # tag, lib, extra args
  rm -rf /tmp/p_x; T2D_LIB_PATH=$2 timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -- python $R/tools/env_only_bench.py --n $N --steps 1000 $3 > /dev/null 2>&1
  echo "N=$N $1: $(python $R/tools/summarize_prof.py stats /tmp/p_x | grep 'k_step2\|k_env' | head -1 | awk '{print $(NF-1)}') us"
}
run base "" ""
run base_noobs "" "--no-obs"
for x in "$@"; do run exp$x $R/scratch_exp/libexp$x.so ""; done
