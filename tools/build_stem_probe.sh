#!/bin/bash
# scratch_exp/libstemprobe.so: the library with csrc/stem_hip.hip compiled -DSTEM_PROBE (the other objects as built by build.py)
#   bash tools/build_stem_probe.sh [output name] [extra compiler flags ...]
out=${1:-libstemprobe.so}; shift
cd "$(dirname "$0")/../active_tracking_rl_amd/csrc" && mkdir -p ../../scratch_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DSTEM_PROBE "$@" -c stem_hip.hip -o /tmp/stem_probe.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -ldl -o ../../scratch_exp/$out /tmp/stem_probe.o $(ls _obj/*.o | grep -v stem_hip.o)
ls -la ../../scratch_exp/$out
