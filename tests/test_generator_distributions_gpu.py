"""-m gpu: the DEVICE generators (Philox + keyed Feistel subset, k_gen / k_gen_nav) pinned to the reference's distributions:
2 x 16 384 generated episodes per case, the statistics of tests/dist_stats.py against tests/golden/distributions.npz (>= 20 000
resets of the reference env per case). See tests/test_generator_distributions_cpu.py for what is compared and why equality
can only be distributional. Reference: generators.py:38-94,115-176; navigator.py:43-63,73-93."""
import numpy as np
import pytest

from dist_stats import collect, compare
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def device_stats(map_type, mode, n, seeds):
    from active_tracking_rl_amd.vec_env import VecTrack2D
    parts = []
    for seed in seeds:
        env = VecTrack2D("Track2D-%sPartial%s-v0" % (map_type, mode), num_envs=n, seed=seed, env_id_base=seed * 1000003)
        env.reset()
        S = 82 if map_type == "Block" else 81
        maps = env.get_maps()[:, :S, :S]
        st = env.get_state()
        tg = env.get_target()
        assert env.faults() == 0
        pos, goals = st["pos"].astype(np.int64), st["goals"].astype(np.int64)
        plan_len = first_act = plan_b = redrawn = path_len = None
        if mode == "Ram":
            plan_len, first_act = tg["len"].astype(np.int64), tg["plan"][:, 0].astype(np.int64)
        if mode == "Nav":
            plan_b = tg["len"] == 10                                   # plan B = 10 random actions (navigator.py:58-59)
            redrawn = (tg["navgoal"] != goals[:, 1]).any(1)
            path_len, first_act = np.zeros(n, np.int64), np.zeros(n, np.int64)
            for i in range(n):                                         # BFS distance target spawn -> Navigator's goal, host side
                if plan_b[i]:
                    continue
                d, dist = orc.bfs_field(maps[i], tg["navgoal"][i])
                path_len[i] = dist[pos[i, 1, 0], pos[i, 1, 1]]
                first_act[i] = d[pos[i, 1, 0], pos[i, 1, 1]]
        parts.append(collect(map_type, mode, maps, pos, goals, plan_len, first_act, plan_b, redrawn, path_len))
        env.close()
    tot = {k: sum(p[k] for p in parts) for k in parts[0]}
    return tot


@pytest.mark.parametrize("case", ["Block_Ram", "Maze_Ram", "Block_Nav", "Maze_Nav"])
def test_device_generators_draw_the_references_distributions(case):
    map_type, mode = case.split("_")
    st = device_stats(map_type, mode, 16384 if mode == "Ram" else 8192, seeds=(3, 4))
    keys = None
    if mode == "Nav":      # (first actions: BFS tie-break vs heap A*, see the CPU twin of this test)
        keys = ["walls", "wall_rows", "wall_cols", "offs", "tr_rows", "tr_cols", "g0_rows", "g1_cols", "plan_len"]
    report = []
    compare(case, st, keys, report)
    print(case, ["%s %.1f/%.1f" % (k, x, lim) for k, x, lim, _ in report])
