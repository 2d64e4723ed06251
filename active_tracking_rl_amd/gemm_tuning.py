"""Library-GEMM algorithm selection for the policy's plain GEMMs (fc 512/1024->256, LSTM projections, heads).

hipBLASLt's default heuristic is far from the best kernel for these skinny fp32 shapes on MI355X (e.g. the target's
fc [81920,1024]x[1024,256]: 0.76 ms by default, 0.30 ms with the rocBLAS kernel the tuner picks). PyTorch's TunableOp
does the search; `tunableop_gfx950.csv` next to this file holds the picks for the BASELINE shapes (4096 envs/GPU,
20-step rollouts), produced on an MI355X by `python tools/tune_gemms.py`. enable() switches TunableOp on in read-only
mode: listed shapes use the recorded kernel, anything else (or a file whose validators do not match the installed
ROCm/hipBLASLt) falls back to the default path. No tuning happens at run time unless tune=True."""
import os

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"on": False}


def enable(tune=False, filename=None):
    if _state["on"] and not tune:
        return True
    if not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return False
    path = filename or RESULTS
    if not tune and not os.path.exists(path):
        return False
    if os.environ.get("ATR_DISABLE_GEMM_TUNING") == "1":
        return False
    try:
        t = torch.cuda.tunable
        t.enable(True)
        t.tuning_enable(bool(tune))
        t.set_filename(path, insert_device_ordinal=False)
        if tune:
            t.set_max_tuning_duration(100)
            t.set_max_tuning_iterations(30)
        elif os.path.exists(path):
            t.read_file(path)
        _state["on"] = True
        return True
    except Exception:
        return False
