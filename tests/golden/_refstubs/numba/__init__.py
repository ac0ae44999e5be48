"""Pass-through stand-in for numba, used ONLY by tests/golden/make_golden.py in the build container
to import the (pure Python) reference as a fixture generator.  Not shipped, not imported by the product."""


def _passthrough(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def deco(fn):
        return fn

    return deco


jit = njit = _passthrough


def vectorize(*dargs, **dkwargs):
    import numpy as np

    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return np.vectorize(dargs[0])

    def deco(fn):
        return fn  # the decorated functions in the reference are already numpy-broadcastable

    return deco


class _T:
    def __call__(self, *a, **k):
        return self


float64 = float32 = complex128 = complex64 = int64 = int32 = _T()
