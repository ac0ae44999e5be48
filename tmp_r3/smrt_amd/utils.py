"""dB helpers (smrt/utils/__init__.py:13-36)."""
import numpy as np


def dB(x):
    return 10 * np.log10(np.maximum(x, 1e-20))


def invdB(x):
    return 10.0 ** (np.asarray(x) / 10.0)
