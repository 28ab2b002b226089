import numpy as np


def flatdim(space):
    from . import Box, Dict, Discrete
    if isinstance(space, Box):
        return int(np.prod(space.shape))
    if isinstance(space, Discrete):
        return space.n
    if isinstance(space, Dict):
        return sum(flatdim(s) for s in space.values())
    raise NotImplementedError


def flatten_space(space):
    from . import Box
    return Box(-np.inf, np.inf, shape=(flatdim(space),))


def flatten(space, x):
    from . import Box, Dict
    if isinstance(space, Box):
        return np.asarray(x, dtype=np.float32).flatten()
    if isinstance(space, Dict):
        return np.concatenate([flatten(s, x[k]) for k, s in space.items()])
    raise NotImplementedError
