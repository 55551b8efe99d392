"""One-off stress: the randomised parity test of tests/test_gpu_fuzz.py on many more random shapes/seeds."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_fuzz as F
from oracle import oracle as O

class Skip(Exception):
    pass
import pytest
n_ok = n_skip = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for it in range(N):
    shape = (int(rng.integers(1, 200)), int(rng.integers(2, 260)))
    if rng.random() < 0.2:
        shape = (int(rng.choice([63, 64, 65, 127, 128, 129, 511, 512, 513])), int(rng.choice([63, 64, 65, 128, 130, 257])))
    F.SHAPES = [shape]
    try:
        F.test_random_rasters(None, O, it * len(F.SHAPES))
        n_ok += 1
    except pytest.skip.Exception:
        n_skip += 1
    except Exception as exc:
        print("FAIL at iteration", it, "shape", shape, type(exc).__name__, str(exc)[:300], flush=True)
        raise
print("stress fuzz:", n_ok, "ok,", n_skip, "skipped")
