"""Identity-decorator stand-in for numba, ours, used ONLY by oracle/gen_golden.py.

The reference's own test-suite runs with NUMBA_DISABLE_JIT=1 (reference
tests/conftest.py:7), i.e. its kernels execute as interpreted Python.  numba is not
installed in the build container, so this shim reproduces exactly that mode: every
decorator returns the undecorated function.  Nothing here is reference code.
"""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(func):
        return func

    return wrap


njit = _identity_decorator
jit = _identity_decorator


def vectorize(*args, **kwargs):
    import numpy as np

    if len(args) == 1 and callable(args[0]) and not kwargs:
        return np.vectorize(args[0])

    def wrap(func):
        return np.vectorize(func)

    return wrap


from . import typed  # noqa: E402,F401
