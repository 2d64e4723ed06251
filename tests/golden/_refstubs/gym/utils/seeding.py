import numpy as np


def np_random(seed=None):
    return np.random.RandomState(seed if seed is not None else 0), seed
