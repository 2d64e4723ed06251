# MFMA-side counters of the policy kernels in situ (one replayed A3C iteration x 20), in their own --pmc passes (no trace
# domains): wave cycles, wait cycles, MFMA busy cycles and MFMA op counts per kernel -> gpurun_out/<tag>/mfma_pmc.txt
TAG=${1:-r02}; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  rm -rf /tmp/p_m; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/p_m -- python $R/tools/iter_profile.py 20 > /dev/null 2>&1
  python - "$set" <<EOF2
import csv, glob, collections, sys
f = glob.glob("/tmp/p_m/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for tag in ("k_actor_step", "k_gemm_tn<", "k_stem_fwd", "k_stem_bwd", "k_step2"):
        if tag in k:
            d[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# pass:", sys.argv[1])
for tag, cs in sorted(d.items()):
    print("%-14s" % tag, "  ".join("%s mean=%.0f (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
EOF2
done > $O/mfma_pmc.txt 2>&1
cat $O/mfma_pmc.txt
