#!/bin/bash
# tools/multirank_probe.py once per hardware-queue setting (read by the HIP runtime at start-up) -> gpurun_out/r04_multirank_1gpu.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export PYTHONPATH=$R; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/r04_multirank_1gpu.txt
for q in default 2 3; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29547 \
      tools/multirank_probe.py 512,1024 2>&1 | grep "^queues" >> $O/r04_multirank_1gpu.txt
done
cat $O/r04_multirank_1gpu.txt
