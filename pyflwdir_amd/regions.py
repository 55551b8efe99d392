"""``pyflwdir.regions`` sums (reference pyflwdir/regions.py:16-57): per-label sums over a label raster.  The
reference delegates them to ``scipy.ndimage`` on the host — they hold no flow-direction logic and are not part
of the device path; here they are the same arithmetic in plain numpy (``np.bincount`` with float64 weights is
what ``ndimage.sum`` evaluates for a list of labels), so that a user of the reference finds the names."""
from __future__ import annotations

import numpy as np

from . import gis

__all__ = ["region_sum", "region_area"]


def region_sum(data, regions):
    """(unique positive region IDs, sum of ``data`` per ID); reference pyflwdir/regions.py:16-32."""
    data, regions = np.asarray(data), np.asarray(regions)
    if data.shape != regions.shape:
        raise ValueError("data and regions must have the same shape")
    lbs, inv = np.unique(regions[regions > 0], return_inverse=True)
    sums = np.bincount(inv, weights=data[regions > 0].astype(np.float64, copy=False), minlength=lbs.size)
    return lbs, sums


def region_area(regions, transform=gis.IDENTITY, latlon=False):
    """(unique region IDs, area [m2] per ID); reference pyflwdir/regions.py:35-57."""
    area = gis.area_grid(transform=transform, shape=np.asarray(regions).shape, latlon=latlon)
    return region_sum(area, regions)
