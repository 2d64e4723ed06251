"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/track2d.h declares; the env-id table equals the reference registry (golden registry.npz); the product
path refuses to run without the GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _header_symbols(header="track2d.h", prefix="t2d_"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, txt)))


def test_library_exports_every_declared_symbol():
    from active_tracking_rl_amd import build, vec_env
    build.build()
    lib = ctypes.CDLL(vec_env.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(vec_env.ABI_SYMBOLS) == syms
    lib.t2d_abi_version.restype = ctypes.c_int
    assert lib.t2d_abi_version() == vec_env.ABI_VERSION
    lib.t2d_config_size.restype = ctypes.c_int
    assert lib.t2d_config_size() == ctypes.sizeof(vec_env._Config) == 64      # action_type sits in former padding
    # the other two headers of the boundary: policy-side kernels and the reference-exact episode source
    from active_tracking_rl_amd import np_mode
    np_syms = _header_symbols("track2d_np.h", "t2d_np_")
    assert sorted(np_mode.NP_SYMBOLS) == np_syms and len(np_syms) >= 9
    atr_syms = _header_symbols("atr_policy.h", "atr_")
    assert len(atr_syms) >= 16 and "atr_stem_forward_u8" in atr_syms
    for s in np_syms + atr_syms:
        assert hasattr(lib, s), s


def test_registry_matches_reference():
    from active_tracking_rl_amd import registry
    rows = np.load(os.path.join(GOLDEN, "registry.npz"))["rows"]
    assert len(rows) == 72 == len(registry.REGISTRY)
    for env_id, mp, ob, lvl, tgt, mx in rows:
        r = registry.REGISTRY[str(env_id)]
        assert (r["map_type"], r["obs_type"], str(r["level"]), r["target_mode"], str(r["max_episode_steps"])) == \
            (str(mp), str(ob), str(lvl), str(tgt), str(mx))
    assert registry.spec("Track2D-BlockPartialPZR-v0")["target_mode"] == "PZR"
    assert registry.spec("Track2D-BlockFullPZR-v0")["obs_type"] == "Full"
    assert registry.spec("Track2D-BlockPartialRPF-v0")["target_mode"] == "RPF"
    for env_id in registry.REGISTRY:                       # all 72 ids resolve
        assert registry.spec(env_id)["max_episode_steps"] == 500
    with pytest.raises(KeyError):
        registry.spec("Track2D-Nope-v0")


def test_no_cpu_fallback():
    import torch
    from active_tracking_rl_amd import vec_env
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vec_env.T2DError):
        vec_env.VecTrack2D("Track2D-BlockPartialPZR-v0", num_envs=4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "active_tracking_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libtrack2d_oracle" not in txt, f


def test_evaluator_train_mode_schedule_follows_the_reference():
    """test.py:84-92 restated (active_tracking_rl_amd.test.schedule_train_modes): expected values derived by hand from the
    reference's branch order — below --init-step: 0; --train-mode 2 and more than iter_th iterations since the last flip:
    1 - mode, then iter_th = --init-step if the new mode is 0 else --adv-step; otherwise --train-mode."""
    import argparse
    from active_tracking_rl_amd.test import schedule_train_modes
    args = argparse.Namespace(init_step=10, adv_step=5, train_mode=2)
    modes, st = [0, 0], {}
    seen = []
    for n_iter in (5, 12, 13, 18, 20, 24):
        schedule_train_modes(args, modes, n_iter, st)
        seen.append(list(modes))
    # 5 < 10 -> 0 | 12 - 0 > 10: rank 0 flips 0 -> 1 (last_iter = 12, iter_th = 5), rank 1 then sees 12 - 12 > 5 false -> 2
    # 13: 1 > 5 false -> 2 | 18: 6 > 5: rank 0 flips 2 -> -1, rank 1 -> 2 | 20 -> 2 | 24: rank 0 flips 2 -> -1
    assert seen == [[0, 0], [1, 2], [2, 2], [-1, 2], [2, 2], [-1, 2]]
    # the ordinary schedule (--init-step warm-up, then --train-mode)
    args = argparse.Namespace(init_step=10, adv_step=None, train_mode=-1)
    modes, st = [0], {}
    assert [list(schedule_train_modes(args, modes, n, st)) for n in (0, 9, 10, 500)] == [[0], [0], [-1], [-1]]
    # --train-mode 2 without --adv-step: the reference stops with AttributeError at its first flip away from the tracker
    args = argparse.Namespace(init_step=3, train_mode=2)
    with pytest.raises(AttributeError):
        schedule_train_modes(args, [0], 7, {})


def test_cooperative_step_shape_limits_match_the_kernel():
    """fused.coop_step_supported (what model._act_step asks before it takes the two-launch step) restates the limits
    atr_coop_env_step enforces (csrc/coop_gemm.h: 16 tiles per layer and 8 env pairs per workgroup, whole 16-row tiles per XCD)."""
    from active_tracking_rl_amd import fused
    ok = lambda n, wg: fused.coop_step_supported(n, 256, 128, wg)
    assert ok(512, 256) and ok(1024, 256) and ok(2048, 256) and ok(128, 256)
    assert not ok(4096, 256)            # 32 gate tiles per workgroup
    assert ok(512, 128) and ok(1024, 128) and not ok(2048, 128)
    assert not ok(500, 256) and not ok(512, 100) and not ok(512, 4)
    assert not fused.coop_step_supported(512, 250, 128, 256) and not fused.coop_step_supported(512, 256, 64, 256)
