"""Minimal 2-D affine transform used when the optional ``affine`` package is absent.

The reference takes an ``affine.Affine`` for ``transform`` (pyflwdir/pyflwdir.py:5,
:331-336, pyflwdir/gis_utils.py:7,13).  Only the operations the hot path touches are
provided: 6-coefficient construction, indexing / iteration, ``*`` with a point pair
or another transform, ``~`` (inverse), ``translation`` / ``scale`` / ``identity``.
If the real package is importable it is used instead (see ``get_affine``).
"""
from __future__ import annotations

from collections import namedtuple

_Base = namedtuple("Affine", "a b c d e f")


class Affine(_Base):
    __slots__ = ()

    def __new__(cls, a, b, c, d, e, f, *rest):
        if rest and tuple(rest) != (0.0, 0.0, 1.0):
            raise TypeError("Affine expects 6 coefficients (or 9 with a 0,0,1 last row)")
        return _Base.__new__(cls, float(a), float(b), float(c), float(d), float(e), float(f))

    # constructors -----------------------------------------------------------------
    @classmethod
    def identity(cls):
        return cls(1.0, 0.0, 0.0, 0.0, 1.0, 0.0)

    @classmethod
    def translation(cls, xoff, yoff):
        return cls(1.0, 0.0, xoff, 0.0, 1.0, yoff)

    @classmethod
    def scale(cls, sx, sy=None):
        if sy is None:
            sy = sx
        return cls(sx, 0.0, 0.0, 0.0, sy, 0.0)

    # properties -------------------------------------------------------------------
    @property
    def xoff(self):
        return self.c

    @property
    def yoff(self):
        return self.f

    @property
    def determinant(self):
        return self.a * self.e - self.b * self.d

    # algebra ----------------------------------------------------------------------
    def __mul__(self, other):
        if isinstance(other, (Affine,)) or (
            hasattr(other, "a") and hasattr(other, "f") and hasattr(other, "__len__") and len(other) >= 6
        ):
            oa, ob, oc, od, oe, of = tuple(other)[:6]
            sa, sb, sc, sd, se, sf = self
            return Affine(
                sa * oa + sb * od,
                sa * ob + sb * oe,
                sa * oc + sb * of + sc,
                sd * oa + se * od,
                sd * ob + se * oe,
                sd * oc + se * of + sf,
            )
        try:
            vx, vy = other
        except (TypeError, ValueError):
            return NotImplemented
        sa, sb, sc, sd, se, sf = self
        return (vx * sa + vy * sb + sc, vx * sd + vy * se + sf)

    def __rmul__(self, other):  # pragma: no cover - mirrors affine's behaviour
        return NotImplemented

    def __invert__(self):
        det = self.determinant
        if det == 0:
            raise ValueError("The transform is not invertible")
        idet = 1.0 / det
        sa, sb, sc, sd, se, sf = self
        ra = se * idet
        rb = -sb * idet
        rd = -sd * idet
        re = sa * idet
        return Affine(ra, rb, -sc * ra - sf * rb, rd, re, -sc * rd - sf * re)


def get_affine():
    """Return the ``Affine`` class to use: the real package when present, else ours."""
    try:  # pragma: no cover - not installed in the build image
        from affine import Affine as _A

        return _A
    except Exception:
        return Affine
