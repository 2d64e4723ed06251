# A/B kernel stats of bench.py with byte vs float observations
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for f in u8 f32; do
  rm -rf /tmp/p_ab; flag=""; [ $f = f32 ] && flag="--f32-obs"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 20 $flag > /dev/null 2>&1
  python $R/tools/summarize_prof.py stats /tmp/p_ab > $O/ab_$f.txt
done
