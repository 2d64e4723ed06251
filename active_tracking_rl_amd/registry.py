"""Env-id table: restates the registry loop of the reference (envs/gym-track2d/gym_track2d/__init__.py:3-18):
72 ids `Track2D-{Maze,Block,Empty}{Full,Partial}{Adv,PZR,Far,Nav,Ram,RPF}-v{0,1}`, each with
max_episode_steps=500. All 72 are built (`RPF` = the Navigator patrolling the four static goal cells,
generators.py:12-19)."""

MAP_TYPES = ("Maze", "Block", "Empty")
OBS_TYPES = ("Full", "Partial")
TARGET_MODES = ("Adv", "PZR", "Far", "Nav", "Ram", "RPF")
MAX_EPISODE_STEPS = 500

MAP_CODE = {"Block": 0, "Maze": 1, "Empty": 2}
TARGET_CODE = {"Adv": 0, "PZR": 1, "Far": 2, "Nav": 3, "Ram": 4, "RPF": 5,
               "Ext": 6}   # not an id: target action supplied by the caller (include/track2d.h T2D_TGT_EXT)

REGISTRY = {}
for _m in MAP_TYPES:
    for _o in OBS_TYPES:
        for _t in TARGET_MODES:
            for _l in range(2):
                REGISTRY["Track2D-%s%s%s-v%d" % (_m, _o, _t, _l)] = dict(
                    map_type=_m, obs_type=_o, level=_l, target_mode=_t, max_episode_steps=MAX_EPISODE_STEPS)


def spec(env_id):
    if env_id not in REGISTRY:
        raise KeyError("unknown env id %r (72 Track2D ids are registered)" % (env_id,))
    return dict(REGISTRY[env_id])
