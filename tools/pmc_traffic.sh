# HBM traffic of the step kernel from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not
# fit one pass), no trace domains besides the counter collection. Usage: bash tools/pmc_traffic.sh <tag> <N>...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_pmc -- python $R/tools/env_only_bench.py --n $n --steps 200 --warmup 20 > /dev/null 2>&1
    python $R/tools/summarize_prof.py pmc /tmp/p_pmc k_step2 > $O/env_only_pmc_${c}_$n.txt
    cat $O/env_only_pmc_${c}_$n.txt
  done
done
