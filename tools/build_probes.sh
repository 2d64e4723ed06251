#!/bin/bash
# Probe builds of the library (compile-time experiments, never the product): scratch_exp/libexp{5,6,7,8}.so (T2D_EXP latency
# probes of the step kernel, tools/exp_variants.sh / timeline_probe.py) and libnomaze.so (generator without the maze-growth
# loop, tools/exp_nav.sh). Run in the build container; the .so files travel to the GPU box with the tree.
cd "$(dirname "$0")/.." && mkdir -p scratch_exp
SRC="track2d_hip.hip stem_hip.hip policy_hip.hip lstm_hip.hip heads_hip.hip gemm_tn_hip.hip actor_step_hip.hip pair_gemm_hip.hip bptt_hip.hip driver_hip.hip gate_cell_hip.hip np_mode.cpp lt_gemm.cpp"
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -ldl"
cd active_tracking_rl_amd/csrc
for x in 5 6 7 8 9; do /opt/rocm/bin/hipcc $FL -DT2D_EXP=$x -o ../../scratch_exp/libexp$x.so $SRC & done
/opt/rocm/bin/hipcc $FL -DT2D_EXP_NOMAZE -o ../../scratch_exp/libnomaze.so $SRC &
/opt/rocm/bin/hipcc $FL -DATR_EXP=1 -o ../../scratch_exp/libatrexp1.so $SRC &
/opt/rocm/bin/hipcc $FL -DATR_TN_PROBE=1 -o ../../scratch_exp/libtnprobe.so $SRC &     # tools/gemm_tn_timeline.py
wait
ls -la ../../scratch_exp/*.so
