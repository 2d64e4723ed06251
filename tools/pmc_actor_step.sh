export PYTHONPATH=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_as2
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES --output-format csv -d /tmp/p_as2 -- python $GRAFT_REPO_ROOT/tools/actor_step_bench.py > /dev/null 2>&1
python - <<EOF2
import csv,glob,collections
f=glob.glob("/tmp/p_as2/**/*counter_collection.csv",recursive=True)[0]
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_actor_step" in r["Kernel_Name"]:
        d[r.get("Grid_Size")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,cs in sorted(d.items(), key=lambda kv:int(kv[0])):
    print("grid", g, {k: round(sum(v)/len(v)) for k,v in cs.items()})
EOF2
