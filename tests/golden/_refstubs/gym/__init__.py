"""Minimal stand-in for gym==0.12.5, used ONLY by tests/golden/make_golden.py in the build
container to import the reference env core (envs/gym-track2d) and capture golden vectors.
Never shipped, never imported by the product or by tests at run time."""
from . import spaces, utils, envs  # noqa: F401


class Env(object):
    metadata = {}

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space
