R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
run() { rm -rf /tmp/p_x; T2D_LIB_PATH=$2 timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_x -- python $R/tools/env_only_bench.py --n $N --steps 1000 > /dev/null 2>&1
  echo "N=$N $1: $(python $R/tools/summarize_prof.py stats /tmp/p_x | grep 'k_step2' | head -1 | awk '{print $(NF-1)}') us"; }
for N in 4096 65536; do run waves4 ""; for w in 1 2 8 16; do run waves$w $R/scratch_exp/libw$w.so; done; done
