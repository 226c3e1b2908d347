"""Host-side mirrors of lightkurve.correctors for the regression hot path."""
from .designmatrix import (DesignMatrix, DesignMatrixCollection, SparseDesignMatrix,  # noqa: F401
                           SparseDesignMatrixCollection, create_sparse_spline_matrix)
from .pldcorrector import PixelCube, PLDCorrector, pld_correct_batch  # noqa: F401
from .regressioncorrector import RegressionCorrector  # noqa: F401
from .metrics import overfit_metric_lombscargle  # noqa: F401
from .cbvcorrector import CBVCorrector  # noqa: F401
