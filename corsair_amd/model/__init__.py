"""Model registry mirroring the reference's model/__init__.py:35-48 (`load_model(name)` -> class)."""
from . import fc, resunet

_MODELS = {c.__name__: c for c in (resunet.ResUNetBN2, resunet.ResUNetBN2B, resunet.ResUNetBN2C,
                                   resunet.ResUNetBN2D, resunet.ResUNetBN2E, resunet.ResUNetIN2,
                                   resunet.ResUNetIN2B, resunet.ResUNetIN2C, resunet.ResUNetIN2D,
                                   resunet.ResUNetIN2E)}
_HEADS = {c.__name__: c for c in (fc.conv1_max_embedding, fc.conv1_chamfer)}


def load_model(name):
    """Class named `name` (e.g. "ResUNetBN2C", evaluation.py:181)."""
    table = dict(_MODELS)
    table.update(_HEADS)
    if name not in table:
        raise ValueError(f"model {name!r} is not available; known: {sorted(table)}")
    return table[name]
