"""Helpers shared by the oracle-pin tests and the GPU parity tests."""
from __future__ import annotations

import hashlib
import os

import numpy as np

from oracle import golden_inputs as GI
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    h.update(str(a.dtype.str).encode())
    h.update(str(a.shape).encode())
    h.update(a.tobytes())
    return h.hexdigest()


class Case:
    """One golden case: inputs, reference digests and (for small cases) full reference outputs."""

    def __init__(self, name, manifest):
        self.name = name
        self.entry = manifest[name]
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.d8 = np.ascontiguousarray(z["d8"], dtype=np.uint8)
        self.shape = self.d8.shape
        self.n = self.d8.size
        self.full = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
        self.transform = tuple(self.entry["transform"])
        self.latlon = self.entry["latlon"]
        if "elevtn" in z.files:
            self.elevtn_in = z["elevtn"]
        elif self.entry.get("synth"):
            s = self.entry["synth"]
            self.elevtn_in = O.synth_elev_f32(s["nrow"], s["ncol"], seed=s["seed"], tilt=s["tilt"], white=s["white"],
                                              nodata_pct=s["nodata_pct"])
        else:
            self.elevtn_in = None
        self.digests = self.entry["digests"]
        self.dtypes = self.entry["dtypes"]

    def check(self, key, got):
        """Bit-exact comparison with the reference output `key` (dtype, shape and bytes)."""
        got = np.asarray(got)
        assert got.dtype.str == self.dtypes[key], f"{self.name}:{key} dtype {got.dtype.str} != {self.dtypes[key]}"
        if key in self.full:
            exp = self.full[key]
            assert got.shape == exp.shape, f"{self.name}:{key} shape {got.shape} != {exp.shape}"
            if not np.array_equal(got, exp, equal_nan=True):
                bad = np.flatnonzero(got.ravel() != exp.ravel())
                raise AssertionError(f"{self.name}:{key}: {bad.size} mismatches, first at {bad[:5]}: "
                                     f"got {got.ravel()[bad[:5]]} expected {exp.ravel()[bad[:5]]}")
        assert digest(got) == self.digests[key], f"{self.name}:{key} digest mismatch"


def derived_inputs(case: Case, upa: np.ndarray, idxs_pit: np.ndarray):
    """Inputs that depend on the (already verified) upstream cell count."""
    thr = GI.threshold(upa)
    idxs, ids = GI.basin_outlets(upa, idxs_pit)
    elevtn = GI.elevation(case.elevtn_in, upa)
    return dict(thr=thr, mask_upa=upa > thr, mask_rand=GI.random_mask(case.shape), basins_idxs=idxs, basins_ids=ids,
                elevtn=elevtn, drain=upa > thr)
