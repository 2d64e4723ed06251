"""In-tree build of the HIP extension (gfx950 only). `python -m active_tracking_rl_amd.build [--force]`.

Every source is compiled to an object of its own (in parallel; only the ones older than their source or any header are
redone) under csrc/_obj/, then linked into libtrack2d_hip.so next to this file."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libtrack2d_hip.so")
SOURCES = ["track2d_hip.hip", "stem_hip.hip", "policy_hip.hip", "lstm_hip.hip", "heads_hip.hip", "gemm_tn_hip.hip",
           "actor_step_hip.hip", "pair_gemm_hip.hip", "bptt_hip.hip", "driver_hip.hip", "gate_cell_hip.hip", "np_mode.cpp", "lt_gemm.cpp"]
HEADERS = ["t2d_device.h", os.path.join("..", "..", "include", "track2d.h"),
           os.path.join("..", "..", "include", "atr_policy.h"), "atr_sample.h", "atr_cell.h",
           os.path.join("..", "..", "include", "track2d_np.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-ldl"]


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(src, hdr_time):
    o = _obj(src)
    return not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(CSRC, src)), hdr_time)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libtrack2d_hip.so next to this file (hipcc cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest_header()
    todo = [s for s in SOURCES if force or _stale(s, hdr_time)]

    def compile_one(src):
        cmd = [HIPCC] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        list(pool.map(compile_one, todo))
    cmd = [HIPCC] + LDFLAGS + ["-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
