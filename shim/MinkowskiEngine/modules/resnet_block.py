"""Names imported at class-definition time by the reference's model/resnet.py:22,136,151 (a
classifier that no entry script instantiates).  Constructing them is not supported."""
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("MinkowskiEngine.modules.resnet_block is not on the CORSAIR inference path")


class Bottleneck(BasicBlock):
    expansion = 4
