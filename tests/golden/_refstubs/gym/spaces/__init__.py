class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
