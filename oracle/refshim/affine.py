"""``affine`` stand-in for importing the reference in the build container (the real
package is not installed).  Re-exports this repo's own minimal Affine."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from pyflwdir_amd._affine import Affine  # noqa: E402,F401
