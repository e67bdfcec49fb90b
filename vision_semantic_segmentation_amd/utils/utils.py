"""src/utils/utils.py:68-75 -- the two helpers the hot path uses."""
import numpy as np


def homogenize(x):
    """inhomogeneous -> homogeneous: append a row of ones (utils.py:68-70)."""
    return np.vstack((x, np.ones((1, x.shape[1]))))


def dehomogenize(x):
    """homogeneous -> inhomogeneous: x[:-1] / x[-1], no zero guard (utils.py:73-75)."""
    return x[:-1] / x[-1]
