import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def sub(npz, prefix):
    """The entries of an npz that start with prefix, with the prefix removed."""
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))
