"""In-tree build of the HIP extension (gfx950 only). `python -m active_tracking_rl_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtrack2d_hip.so")
SOURCES = ["track2d_hip.hip", "stem_hip.hip", "policy_hip.hip", "lstm_hip.hip", "heads_hip.hip", "gemm_tn_hip.hip",
           "actor_step_hip.hip", "pair_gemm_hip.hip", "bptt_hip.hip", "driver_hip.hip", "np_mode.cpp"]
HEADERS = ["t2d_device.h", os.path.join("..", "..", "include", "track2d.h"),
           os.path.join("..", "..", "include", "atr_policy.h"), "atr_sample.h", "atr_cell.h",
           os.path.join("..", "..", "include", "track2d_np.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libtrack2d_hip.so next to this file (hipcc cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
