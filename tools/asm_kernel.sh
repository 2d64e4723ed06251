# Device assembly of one kernel of csrc/track2d_hip.hip: tools/asm_kernel.sh <mangled-name-regex> [extra -D flags]
cd "$(dirname "$0")/.." && mkdir -p scratch_exp
pat=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only "$@" -o scratch_exp/t2d.s active_tracking_rl_amd/csrc/track2d_hip.hip 2>&1 | grep -v "warning\|^$" | head
awk -v pat="^$pat:" '$0 ~ pat {f=1} f{print} f && /^\.Lfunc_end/ {exit}' scratch_exp/t2d.s > scratch_exp/kernel.s
wc -l scratch_exp/kernel.s
