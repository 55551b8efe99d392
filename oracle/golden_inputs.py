"""TEST INFRASTRUCTURE — construction of the derived inputs of the golden cases.

Shared by oracle/gen_golden.py (which feeds them to the REFERENCE) and by the tests (which
feed the same inputs to the oracle and to the HIP path), so that the three can never drift.
Everything is deterministic: hash-based weights from oracle.synth_weights_f32, thresholds
derived from the upstream-cell-count raster.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

DEFAULT_TRANSFORM = (1 / 120.0, 0.0, 5.0, 0.0, -1 / 120.0, 50.0)  # lat/lon, 30 arcsec
PROJ_TRANSFORM = (30.0, 0.0, 1000.0, 0.0, -30.0, 5000.0)           # projected, 30 m


def payloads(shape):
    n = int(np.prod(shape))
    w32 = O.synth_weights_f32(n, seed=1).reshape(shape)
    wi = (w32 * 1000).astype(np.int32)
    wi[(wi % 17) == 0] = -9999  # nodata cells inside the domain
    wf = w32.copy()
    wf[(wi % 13) == 0] = -1.0
    return dict(w32=w32, w64=w32.astype(np.float64) * 3.25, wi32_nodata=wi, wf32_nodata_m1=wf,
                wi64=(w32 * 1e6).astype(np.int64) * 100003)


def threshold(upa):
    return max(2, int(np.percentile(upa[upa > 0], 80))) if (upa > 0).any() else 2


def random_mask(shape):
    n = int(np.prod(shape))
    return O.synth_weights_f32(n, seed=7).reshape(shape) < 0.6


def basin_outlets(upa, idxs_pit):
    """Nested outlets: a few pits plus a sample of high-accumulation interior cells; int16 ids."""
    n = upa.size
    order = np.argsort(upa.ravel(), kind="stable")[::-1]
    cand = order[: max(4, min(200, n // 50))]
    sel = cand[:: max(1, cand.size // 23)][:23]
    sel = sel[upa.ravel()[sel] > 0]
    idxs = np.unique(np.concatenate([np.asarray(idxs_pit[: min(5, idxs_pit.size)], dtype=np.int64),
                                     sel.astype(np.int64)]))
    ids = (np.arange(idxs.size, dtype=np.int16) * 7 + 3).astype(np.int16)
    return idxs, ids


def elevation(case_elevtn, upa):
    if case_elevtn is not None:
        return np.ascontiguousarray(case_elevtn, dtype=np.float32)
    n = upa.size
    return (O.synth_weights_f32(n, seed=11).reshape(upa.shape) * 50 + upa.clip(0) ** 0.1).astype(np.float32)
