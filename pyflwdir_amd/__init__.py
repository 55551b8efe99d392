"""pyflwdir_amd — MI355X-native D8 flow-accumulation hot path behind the pyflwdir
``FlwdirRaster`` API (reference: Deltares/pyflwdir v0.5.12).

    import pyflwdir_amd as pyflwdir
    flw = pyflwdir.from_array(d8, ftype="d8", transform=transform, latlon=True)
    upa = flw.upstream_area("km2"); sto = flw.stream_order(); bas = flw.basins()

Python host code -> ctypes -> C-ABI (include/pfd.h) -> hand-written HIP kernels (gfx950).
"""
from . import gis as gis_utils  # reference name of the module
from . import gis
from . import dem
from . import regions
from .nextxy import read_nextxy
from .raster import FTYPES, FlwdirRaster, from_array, from_dem

__version__ = "0.1.0"
__all__ = ["FlwdirRaster", "from_array", "from_dem", "dem", "regions", "read_nextxy", "gis_utils", "gis", "FTYPES"]
