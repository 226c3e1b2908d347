"""lightkurve_amd — MI355X-native (gfx950 / HIP) periodogram + systematics-correction hot path of lightkurve.

The package is a thin host-side mirror of the reference interface for that one path
(``LombScarglePeriodogram/BoxLeastSquaresPeriodogram.from_lightcurve``, ``LightCurve.flatten``,
``RegressionCorrector/PLDCorrector.correct``) over a C-ABI shared library of hand-written HIP kernels
(``lightkurve_amd/csrc`` -> ``liblkhip.so``, declared in ``include/lkhip.h``).  Importing the package
does not load the library; the first compute call does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"

from .lightcurve import FoldedLightCurve, LightCurve  # noqa: F401
from .ingest import LightCurveBatch  # noqa: F401
from .device import DeviceLightCurveBatch  # noqa: F401
from .fitsio import read  # noqa: F401
from .periodogram import (BoxLeastSquaresPeriodogram, LightkurveWarning, LombScarglePeriodogram,  # noqa: F401
                          Periodogram)
