# LDS / issue counters of the stem kernels at the learner's launch size (tools/stem_bench.py 163840), each set in its own --pmc pass
# (no trace domains): where k_stem_bwd16 / k_stem_fwd16 wait  -> gpurun_out/<tag>/stem_lds_pmc.txt
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"; do
  rm -rf /tmp/p_m; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/p_m -- python $R/tools/stem_bench.py 163840 > /dev/null 2>&1
  python - "$set" <<EOF2
import csv, glob, collections, sys
fs = glob.glob("/tmp/p_m/**/*counter_collection.csv", recursive=True)
print("# pass:", sys.argv[1])
if fs:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for tag in ("k_stem_fwd16<unsigned char>", "k_stem_bwd16<unsigned char>"):
            if tag in k:
                d[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for tag, cs in sorted(d.items()):
        print("%-30s" % tag, "  ".join("%s mean=%.0f (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
else:
    print("(no counter file)")
EOF2
done > $O/stem_lds_pmc.txt 2>&1
cat $O/stem_lds_pmc.txt
