"""`import MinkowskiEngine.MinkowskiFunctional as MEF` (model/resunet.py:20, residual_block.py:21)."""
from corsair_amd.minkowski import MinkowskiFunctional as _F

relu = _F.relu
