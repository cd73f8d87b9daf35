"""Import shim: the product package lives in the directory `r-yolov4_amd/` (not a valid Python identifier),
so `import ryolov4_amd` maps onto it.  See r-yolov4_amd/_pkg.py for the real package init."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "r-yolov4_amd")]
from ._pkg import *          # noqa: F401,F403,E402
from ._pkg import install_dropin  # noqa: F401,E402
