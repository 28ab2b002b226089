import builtins

import numpy as np


class Space:
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()

    def sample(self):
        return int(np.random.randint(self.n))


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.dtype = dtype
        self.low = np.full(self.shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(self.shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)


class Dict(Space, builtins.dict):
    def __init__(self, spaces=None, **kw):
        builtins.dict.__init__(self, spaces or {}, **kw)
        self.spaces = self


class Tuple(Space, builtins.tuple):
    def __new__(cls, spaces):
        return builtins.tuple.__new__(cls, spaces)

    def __init__(self, spaces):
        self.spaces = builtins.tuple(spaces)


from . import utils  # noqa: E402,F401
