export PYTHONPATH=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_as
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_as -- python $GRAFT_REPO_ROOT/tools/actor_step_bench.py > /dev/null 2>&1
python - <<EOF2
import csv,glob,collections
f=glob.glob("/tmp/p_as/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_actor_step" in r["Kernel_Name"]:
        d[r.get("Grid_Size_X") or r.get("Grid_Size")].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:int(kv[0])): print("grid threads", k, "calls", len(v), "avg us %.2f"%(sum(v)/len(v)/1e3), "min %.2f"%(min(v)/1e3))
EOF2
