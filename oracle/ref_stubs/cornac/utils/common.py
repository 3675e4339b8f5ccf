import numbers

import numpy as np


def get_rng(seed):
    if seed is None:
        return np.random.mtrand._rand
    if isinstance(seed, (numbers.Integral, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r can not be used to create a numpy.random.RandomState" % (seed,))


def scale(values, target_min, target_max, source_min=None, source_max=None):
    lo = np.min(values) if source_min is None else source_min
    hi = np.max(values) if source_max is None else source_max
    span = (hi - lo) or 1.0
    return (values - lo) * (target_max - target_min) / span + target_min
