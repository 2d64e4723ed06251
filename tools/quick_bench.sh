# bench.py (3 repeats) + the per-iteration kernel table
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 600 python bench.py --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_launch_us'])"
bash tools/prof_iter.sh > /dev/null 2>&1
head -3 $O/iteration_kernel_stats.txt
awk 'NR>3 && NF>4 {n=NF; calls=$(n-3); tot=$(n-2); avg=$(n-1); if (avg<9) {c+=calls; t+=tot} else {C+=calls; T+=tot}} END{print "small(<9us): calls/iter", c/52, "ms/iter", t/52; print "big: calls/iter", C/52, "ms/iter", T/52}' $O/iteration_kernel_stats.txt
