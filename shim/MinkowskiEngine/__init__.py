"""`import MinkowskiEngine as ME` -> the MI355X-native operator surface (corsair_amd.minkowski).

Put this directory's parent (`shim/`) on PYTHONPATH and the reference's model/*.py and evaluation.py
import unchanged.  No MinkowskiEngine code is involved."""
from corsair_amd.minkowski import (  # noqa: F401
    CoordinateManager,
    CoordinateMapKey,
    MinkowskiAvgPooling,
    MinkowskiBatchNorm,
    MinkowskiBroadcastMultiplication,
    MinkowskiConvolution,
    MinkowskiConvolutionTranspose,
    MinkowskiGlobalMaxPooling,
    MinkowskiGlobalPooling,
    MinkowskiInstanceNorm,
    MinkowskiLinear,
    MinkowskiNetwork,
    MinkowskiReLU,
    MinkowskiSumPooling,
    SparseTensor,
    __version__,
    cat,
    utils,
)
from . import MinkowskiFunctional  # noqa: F401
