REGISTRY = {}


def register(id, entry_point=None, kwargs=None, max_episode_steps=None, **kw):
    REGISTRY[id] = dict(entry_point=entry_point, kwargs=kwargs or {}, max_episode_steps=max_episode_steps)
